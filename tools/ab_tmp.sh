set -e
cd $GRAFT_REPO_ROOT
cp recommenders_amd/libifr.so /tmp/libifr_new.so
timeout -k 10 300 python -u -m pytest tests/test_gpu_itemitem.py tests/test_gpu_config_sizes.py -x -q --timeout 300 --timeout-method thread -k "csr or c5" 2>&1 | tail -2
for r in 1 2; do
timeout -k 10 200 python -u tools/time_c5.py 524288 0 0 10,20,32 1 > gpurun_out/ab_new_$r.log 2>&1; grep "^dbg" gpurun_out/ab_new_$r.log | sed "s/^/new pair /"
cp recommenders_amd/libifr_old.so recommenders_amd/libifr.so
timeout -k 10 200 python -u tools/time_c5.py 524288 0 0 10,20,32 1 > gpurun_out/ab_old_$r.log 2>&1; grep "^dbg" gpurun_out/ab_old_$r.log | sed "s/^/old pair /"
cp /tmp/libifr_new.so recommenders_amd/libifr.so
done
