#!/bin/bash
# Counterpart of the reference's run_train.sh (same arguments and environment knobs; body: _launch.sh):  run_train.sh <config.yaml> [driver flags]
DRIVER=train_accum.py DEFAULT_PORT=1235 source "$(dirname "$0")/_launch.sh" "$@"
