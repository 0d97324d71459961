python -m pytest tests -m gpu -x -q 2>&1 | tail -5
tests/profile_c5.sh r05_c5b 500000 2>&1 | tail -60
