#!/bin/bash
# usage: tools/run_with_lib.sh <variant .so> <command...>: runs the command with the variant library in place of libmvdseg_hip.so
set -e
V=$1; shift
cp multimodal_mvd_seg_amd/libmvdseg_hip.so /tmp/libmvdseg_hip.so.keep
cp $V multimodal_mvd_seg_amd/libmvdseg_hip.so
"$@" || true
cp /tmp/libmvdseg_hip.so.keep multimodal_mvd_seg_amd/libmvdseg_hip.so
