#!/bin/bash
# Counterpart of the reference's run_extract_feature.sh (latent shards; body: _launch.sh):  run_extract_feature.sh <config.yaml>
DRIVER=extract_features.py DEFAULT_PORT=1235 source "$(dirname "$0")/_launch.sh" "$@"
