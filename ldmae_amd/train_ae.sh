#!/bin/bash
# Counterpart of the reference's VMAE/train_ae.sh (the tokenizer recipe), for the stages this package builds:
#   Stage 1  VMAE pre-training at 128 x 128 (main_pretrain.py -> vmae_pretrain.py; one process per GPU, RCCL all-reduce)
#   Stage 2  position-embedding reset to the 256 x 256 grid (pe_reset.py)
#   Stage 3  decoder fine-tuning with the LPIPS loss: NOT built (needs the VGG weights; SURVEY section 2.1 #18 marks it out) -- printed, not run.
# Stage 1's flags are the reference script's (train_ae.sh:26-46) minus --perceptual_loss_ratio (the LPIPS term; vmae_pretrain.py refuses it by name rather than
# train a different objective silently); torch.distributed.run replaces the deprecated torch.distributed.launch; fp16 = the reference's torch.amp.autocast('cuda').
GPUS_PER_NODE=${GPUS_PER_NODE:-8}
DATA_PATH=${DATA_PATH:-/data/dataset/imagenet/1K_dataset}
OUT=${OUT:-./work_dir/vmae_before_decoder_finetuning}
export HSA_ENABLE_IPC_MODE_LEGACY=${HSA_ENABLE_IPC_MODE_LEGACY:-0}     # dmabuf IPC: RCCL needs it on this driver
cd "$(dirname "$0")" || exit 1

stage1=(--batch_size 128 --no_cls --accum_iter 2 --num_workers 12 --smooth_output --fixed_std 1e-3 --model mae_for_ldmae_f8d16_prev --input_size 128
        --mask_ratio 0.25 --visible_loss_ratio 0.75 --epochs 400 --warmup_epochs 10 --blr 1.0e-4 --weight_decay 0.05 --kl_loss_weight 1e-6 --precision fp16
        --data_path "$DATA_PATH" --output_dir "$OUT" --log_dir "$OUT")
echo "Stage 1: VMAE pre-training (128 x 128, mask ratio 0.25)"
python -m torch.distributed.run --nproc-per-node "$GPUS_PER_NODE" --nnodes 1 --node-rank 0 --master-addr 127.0.0.1 vmae_pretrain.py "${stage1[@]}" "$@" || exit 1

echo "Stage 2: position embeddings of epoch 90's checkpoint -> the 256 x 256 grid"
python pe_reset.py --model_name mae_for_ldmae_f8d16_prev --ckpt_dir "$OUT/checkpoint-90.pth" || exit 1

echo "Stage 3 (decoder tuning at 256 x 256 with --tune_decoder --perceptual_loss_ratio 10.0) is outside this package: run the reference's"
echo "main_pretrain.py for it on $OUT/checkpoint-90.pth (the checkpoint's 'model' entry is the reference's state dict)."
