# Shared body of the run_*.sh launchers (sourced, not run).  The caller sets
#   DRIVER        the script to start (train_accum.py, inference.py, extract_features.py)
#   DEFAULT_PORT  rendezvous port when MASTER_PORT is unset (the reference scripts' own defaults: 1235 / 1236 / 1237)
#   FIXED_SINGLE  "1": one process on this machine whatever the environment says (run_fast_inference.sh)
#   DRIVER_FLAGS  extra flags for the driver (e.g. --demo)
# and passes the config path as $1; further arguments go to the driver.
# Environment contract of the reference scripts: GPUS_PER_NODE (8), WORLD_SIZE = number of machines (1), RANK = this machine's index (0),
# MASTER_ADDR (127.0.0.1), MASTER_PORT, PRECISION (bf16).  One process per GPU is started with `accelerate launch` when it is installed (the
# reference's launcher) and with torch.distributed.run otherwise, or when LDMAE_USE_TORCHRUN is set; both export RANK / LOCAL_RANK / WORLD_SIZE /
# MASTER_ADDR / MASTER_PORT, which is all the drivers read.
config=$1
shift
if [ "$FIXED_SINGLE" = "1" ]; then
  per_node=1; machines=1; machine=0; accel_cfg=()
else
  per_node=${GPUS_PER_NODE:-8}; machines=${WORLD_SIZE:-1}; machine=${RANK:-0}
  accel_cfg=(--config-file configs/accelerator/8gpu.yaml)
fi
addr=${MASTER_ADDR:-127.0.0.1}
port=${MASTER_PORT:-$DEFAULT_PORT}
export PRECISION=${PRECISION:-bf16}
export HSA_ENABLE_IPC_MODE_LEGACY=${HSA_ENABLE_IPC_MODE_LEGACY:-0}     # dmabuf IPC: RCCL needs it on this driver
echo "$config"
cd "$(dirname "${BASH_SOURCE[0]}")" || exit 1
unset WORLD_SIZE RANK                                                    # they meant machines above; the launcher sets the per-process ones
if command -v accelerate >/dev/null 2>&1 && [ -z "$LDMAE_USE_TORCHRUN" ]; then
  accelerate launch "${accel_cfg[@]}" --main_process_ip "$addr" --main_process_port "$port" --machine_rank "$machine" \
      --num_processes $((per_node * machines)) --num_machines "$machines" --mixed_precision "$PRECISION" \
      "$DRIVER" --config "$config" $DRIVER_FLAGS "$@"
else
  python -m torch.distributed.run --nnodes "$machines" --node-rank "$machine" --nproc-per-node "$per_node" --master-addr "$addr" --master-port "$port" \
      "$DRIVER" --config "$config" $DRIVER_FLAGS "$@"
fi
