#!/bin/bash
# Baselines of the reference's trainer_files/{trainer_multi_task,trainer_fine_tune}.sh on MI355X.
set -e
cd "$(dirname "$0")/.."
GPUS=${GPUS:-8}
RUN="python -m torch.distributed.run --nnodes=1 --nproc-per-node $GPUS --master-addr 127.0.0.1 --master-port ${PORT:-29500} -m"
export HSA_ENABLE_IPC_MODE_LEGACY=0
python -c "import __graft_entry__ as g; g.build()"

echo "----- multi-task joint CS+BDD+IDD -----"
$RUN mdil_ss_amd.train_multi_task --savedir MultiTask/CSBDDIDD --dataset CSBDDIDD --datasets CS BDD IDD \
  --num-classes 20 20 27 --nb_tasks 3 --num-epochs 150 --batch-size 6 "$@"

echo "----- fine-tuning CS -> BDD, then CS|BDD -> IDD (needs a single-task CS checkpoint in \$CS_CKPT) -----"
$RUN mdil_ss_amd.main_ftp1_enc_newbn --savedir Finetune/CS_BDD --finetune --state "$CS_CKPT" \
  --dataset-old cityscapes --dataset-new BDD --num-epochs 150 --batch-size 6 "$@"
$RUN mdil_ss_amd.main_FT2_flexible_new --savedir Finetune/CSBDD_IDD --finetune \
  --state ../save/Finetune/CS_BDD/model_best_erfnet_ftp1_150_6_Finetune-CStoBDD-final.pth.tar \
  --dataset-new IDD --datasets cityscapes BDD IDD --num-classes 20 20 27 --num-epochs 150 --batch-size 6 "$@"
