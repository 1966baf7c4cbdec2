#!/bin/bash
# Counterpart of the reference's run_inference.sh (sampling on every GPU; body: _launch.sh):  run_inference.sh <config.yaml>
DRIVER=inference.py DEFAULT_PORT=1237 source "$(dirname "$0")/_launch.sh" "$@"
