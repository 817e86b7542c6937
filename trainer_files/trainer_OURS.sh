#!/bin/bash
# CS -> BDD -> IDD with the proposed method (RAP + domain-adaptive KD) on one 8xMI355X node:
# the three commands of the reference's trainer_files/trainer_OURS.sh, one process per GPU.
# Dataset roots: --cs-datadir / --bdd-datadir / --idd-datadir (or MDIL_{CS,BDD,IDD}_DATADIR);
# add "--synthetic 512" to run on the seeded procedural dataset instead.
set -e
cd "$(dirname "$0")/.."
GPUS=${GPUS:-8}
RUN="python -m torch.distributed.run --nnodes=1 --nproc-per-node $GPUS --master-addr 127.0.0.1 --master-port ${PORT:-29500} -m"
export HSA_ENABLE_IPC_MODE_LEGACY=0
python -c "import __graft_entry__ as g; g.build()"

echo "----- STEP 1: CS -----"
$RUN mdil_ss_amd.train_RAPFT_step1 --savedir Adaptations/RAP_FT_CS1 --num-epochs 150 --batch-size 6 \
  --num-classes 20 --current_task=0 --dataset=cityscapes "$@"

echo "----- STEP 2: CS -> BDD (KLD, lambdac 0.1) -----"
$RUN mdil_ss_amd.train_new_task_step2 --savedir Adaptations/RAP_FT_KLD/CS1_BDD2 --num-epochs 150 \
  --model-name-suffix=ours-CS1-BDD2 --batch-size 6 \
  --state ../save/Adaptations/RAP_FT_CS1/model_best_cityscapes_erfnet_RA_parallel_150_6RAP_FT_step1.pth.tar \
  --dataset=BDD --dataset_old=cityscapes --num-classes 20 20 --current_task=1 --nb_tasks=2 --num-classes-old 20 "$@"

echo "----- STEP 3: CS|BDD -> IDD -----"
$RUN mdil_ss_amd.train_new_task_step3 --savedir Adaptations/RAP_FT_KLD/CS1_BDD2_IDD3 --num-epochs 150 \
  --model-name-suffix=OURS-CS1-BDD2-IDD3 --batch-size 6 \
  --state ../save/Adaptations/RAP_FT_KLD/CS1_BDD2/checkpoint_BDD_erfnet_RA_parallel_150_6ours-CS1-BDD2_step2.pth.tar \
  --dataset-new=IDD --datasets cityscapes BDD IDD --num-classes 20 20 27 --num-classes-old 20 20 \
  --current_task=2 --nb_tasks=3 --lambdac=0.1 "$@"
