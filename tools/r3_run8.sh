set -e
mkdir -p gpurun_out
python -m pytest tests/test_gpu_rccl.py tests/test_distributed.py tests/test_gpu_bench_contract.py -x -q > gpurun_out/r3_coll_tests.log 2>&1 || { tail -40 gpurun_out/r3_coll_tests.log; exit 1; }
tail -3 gpurun_out/r3_coll_tests.log
for cfg in C3 C4 C5; do bash tools/pmc_sweep.sh r03_pmc_$(echo $cfg | tr A-Z a-z) $cfg 1 > gpurun_out/r03_pmc_$cfg.log 2>&1; tail -4 gpurun_out/r03_pmc_$cfg.log; done
python tools/make_valu_model.py r03 C3=gpurun_out/r03_pmc_c3 C4=gpurun_out/r03_pmc_c4 C5=gpurun_out/r03_pmc_c5 && cp profiles/valu_issue_model.json gpurun_out/r03_valu_issue_model.json && cp profiles/r03_pmc_*_sweep.json gpurun_out/
for w in 8 4 2; do python bench.py --emulate-world $w --steps 50 > gpurun_out/r03_rank_share_w$w.json 2> gpurun_out/r03_rank_share_w$w.err; python -c "
import json; d=json.load(open('gpurun_out/r03_rank_share_w$w.json')); print('world $w', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['roofline'].get('frac_at_sustained_clock'), d['roofline']['sustained_clock'] and d['roofline']['sustained_clock']['mean_mhz'])"; done
