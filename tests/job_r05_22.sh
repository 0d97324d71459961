export AMD_LOG_LEVEL=0
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
tests/profile_round.sh r05h 2>&1 | tail -22
