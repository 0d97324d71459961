#!/bin/bash
# GPU box: the two-lane launch tail against the plain launch, same box, interleaved (tests/ab_lib_env.sh specs)
# usage: tests/ab_tail.sh [lib]      (lib: variant name, default = the shipped library)
lib=${1:-default}
python tests/micro/tail_identity.py 300000 0.1 0 || exit 1
python tests/micro/tail_identity.py 300000 0.07 1 || exit 1
python tests/micro/tail_identity.py 200000 0.2 0 0.3 || exit 1
tests/ab_lib_env.sh $lib:SF_TAIL_FRAC=0 $lib:SF_TAIL_FRAC=0.04 $lib:SF_TAIL_FRAC=0.08 $lib:SF_TAIL_FRAC=0.12 $lib:SF_TAIL_FRAC=0.16 \
   $lib:SF_TAIL_FRAC=0.08,SF_TAIL_POS=1 $lib:SF_TAIL_FRAC=0.16,SF_TAIL_POS=1
