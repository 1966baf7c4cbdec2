#!/bin/bash
# Counterpart of the reference's run_extract_feature.sh: same arguments and environment knobs.  Launches one process per GPU with
# `accelerate launch` when it is installed (the reference's launcher), else with torch.distributed.run; both export
# RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT, which is all extract_features.py reads.
CONFIG_PATH=$1

GPUS_PER_NODE=${GPUS_PER_NODE:-8}
NNODES=${WORLD_SIZE:-1}
NODE_RANK=${RANK:-0}
MASTER_ADDR=${MASTER_ADDR:-127.0.0.1}
MASTER_PORT=${MASTER_PORT:-1235}
export PRECISION=${PRECISION:-bf16}
export HSA_ENABLE_IPC_MODE_LEGACY=${HSA_ENABLE_IPC_MODE_LEGACY:-0}     # dmabuf IPC: RCCL needs it on this driver

echo $CONFIG_PATH
cd "$(dirname "$0")"
unset WORLD_SIZE RANK
if command -v accelerate >/dev/null 2>&1 && [ -z "$LDMAE_USE_TORCHRUN" ]; then
  accelerate launch \
      --config-file configs/accelerator/8gpu.yaml \
      --main_process_ip $MASTER_ADDR \
      --main_process_port $MASTER_PORT \
      --machine_rank $NODE_RANK \
      --num_processes $(($GPUS_PER_NODE*$NNODES)) \
      --num_machines $NNODES \
      --mixed_precision $PRECISION \
      extract_features.py \
      --config $CONFIG_PATH "${@:2}"
else
  python -m torch.distributed.run --nnodes $NNODES --node-rank $NODE_RANK --nproc-per-node $GPUS_PER_NODE \
      --master-addr $MASTER_ADDR --master-port $MASTER_PORT \
      extract_features.py --config $CONFIG_PATH "${@:2}"
fi
