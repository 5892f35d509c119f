#!/bin/bash
# Round 6, VERDICT item 5: STEP-level A/B of every switchable kernel / engine variant of the final tree, with the columns that
# matter under the power cap: frames/s, socket W, shader MHz, JOULES PER FRAME (SMI energy accumulator over the timed region) and
# the PPT (package power tracking) throttler's residency.  Three interleaved pairs each.  -> profiles/r06_energy_ab.txt
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
ab() { echo "## $1   vs   $2"; bash tools/exp_ab.sh "$1" "$2" 3; }
ab "--precision f16x3" "--precision f16x2"
ab "TA_CONV_NO_WIN=1" "TA_CONV_NO_WIN="
ab "TA_CONV_NO_W2=1" "TA_CONV_NO_W2="
ab "TA_CONV_NO_FASTDRAIN=1" "TA_CONV_NO_FASTDRAIN="
ab "--inflight 2" "--inflight 4"
ab "--embed-min-crops 128 --embed-max-crops 192" "--embed-min-crops 320 --embed-max-crops 512"
ab "--precision f32" "--precision bf16"
