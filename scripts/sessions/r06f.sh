#!/bin/bash
. scripts/ab_lib.sh r06f
timeout 600 python scripts/warm_weights.py 2> $O/warm.err | tee $O/warm_weights.txt; tail -2 $O/warm.err
