set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03d; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log; tail -5 $O/pytest.log
python bench.py --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; echo "bench rc=$?"
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 20 --warmup 5 --gather-leg --no-configs --no-cpu-baseline > $O/bench_tdr_gather.json 2> $O/bench_tdr_gather.err; echo "tdr rc=$?"
# one --pmc pass of the driver command with the program directly after `--` and NO SSD_AQL_SYNC: the library must pick host-side waits itself
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES --output-format csv -d $O/pmc_auto -- python3 bench.py --steps 20 --warmup 5 --no-extras > $O/pmc_auto.log 2>&1; echo "pmc rc=$?"
grep '^{' $O/pmc_auto.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('under --pmc:', d['ms_per_step']*1e3, 'us per step;', d['config']['dispatch'])"
rm -rf $O/pmc_auto
for S in 0 1 2 3; do SSD_AQL_VERBOSE=1 timeout -k 5 120 python tools/queue_probe.py $S 0 2>&1 | grep -v amdgpu.ids >> $O/queue_probe_product.txt; done
cat $O/queue_probe_product.txt | cut -c1-330
