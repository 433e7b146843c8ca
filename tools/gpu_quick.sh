#!/bin/bash
# One short gpurun call: TESTS="pytest args" [EXTRA="cmd"] tools/gpu_quick.sh TAG
TAG=${1:-q}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
if [ -n "$TESTS" ]; then ( time timeout 1200 python -m pytest $TESTS -q -x 2>&1 ) > $OUT/pytest.log 2>&1; tail -40 $OUT/pytest.log; fi
if [ -n "$EXTRA" ]; then ( eval "$EXTRA" ) > $OUT/extra.log 2>&1; tail -40 $OUT/extra.log; fi
echo done
