#!/bin/bash
# same-box A/B of two libraries on the batched Lanczos cases (tools/lab/ablate/time_one.py), three interleaved passes
for p in 1 2 3; do for L in "$@"; do timeout 300 python tools/lab/ablate/time_one.py $L 2>&1 | grep ablate; done; done
