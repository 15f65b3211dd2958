#!/bin/bash
# Phase profile of the MC kernel with in-kernel cycle counters (diagnostic build, not shipped).
set -e
cd "$(dirname "$0")/.."
LBMPC_LIB=$PWD/learning-based-mpc_amd/csrc/prof_lbmpc.so timeout -k 10 200 python tools/prof_phases.py
