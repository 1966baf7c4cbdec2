#!/bin/bash
# Counterpart of the reference's run_fast_inference.sh (one process, the --demo sheet; body: _launch.sh):  run_fast_inference.sh <config.yaml>
DRIVER=inference.py DEFAULT_PORT=1236 FIXED_SINGLE=1 DRIVER_FLAGS=--demo source "$(dirname "$0")/_launch.sh" "$@"
