#!/bin/bash
# Local helper (this container): run a command on the GPU box with the round's scratch directory in place.
# usage: bash tools/gpu.sh <log name> <timeout s> '<command>'
NAME=$1; TMO=$2; shift 2
mkdir -p gpurun_out/r6
/usr/local/graft/bin/gpurun --timeout $TMO -- "mkdir -p gpurun_out/r6 && $*" > gpurun_out/r6/${NAME}_call.log 2>&1
echo "exit $?" >> gpurun_out/r6/${NAME}_call.log
