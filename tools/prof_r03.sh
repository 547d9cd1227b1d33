#!/bin/bash
# Round-3 profiles of the four bench workloads (kernel trace + PMC passes each): bash tools/prof_r03.sh
cd $GRAFT_REPO_ROOT
bash tools/prof.sh r03_c5 item_sparse_topk --config c5_rp3_ebnerd --steps 3 --warmup 1 || exit 1
bash tools/prof.sh r03_c2 score_topk_kernel --steps 5 --warmup 1 || exit 2
bash tools/prof.sh r03_c4 item_item_kernel --config c4_ease_mind --steps 3 --warmup 1 || exit 3
bash tools/prof.sh r03_c3 score_topk_kernel --config c3_mf_shard --steps 2 --warmup 1 || exit 4
echo all-done
