# where the waves' cycles go (MI355X guide, SQ counters: WAIT_ANY + WAIT_INST_ANY + ACTIVE_INST_ANY ~ WAVE_CYCLES), per kernel, cfg3
export TMPDIR=/tmp
cfg=${1:-cfg3}
rm -rf gpurun_out/pmc_sq1 gpurun_out/pmc_sq2
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS --output-format csv -d gpurun_out/pmc_sq1 -- python3 bench.py --config $cfg --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_sq1.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_MFMA SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d gpurun_out/pmc_sq2 -- python3 bench.py --config $cfg --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_sq2.log 2>&1 || exit 1
python3 - <<'PY'
import csv,glob,collections
for d in ("gpurun_out/pmc_sq1","gpurun_out/pmc_sq2"):
    acc=collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(d+"/**/*counter_collection.csv",recursive=True):
        for r in csv.DictReader(open(f)):
            k=r["Kernel_Name"]
            for key in ("fused_fwd2_kernel","lstm_bwd_rs_stream_kernel","gemm_nt_big_kernel"):
                if key in k:
                    tag=key+("<46,0,1>" if "ELi0ELi1E" in k or "<46, 0, 1>" in k else "")
                    acc[tag][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in acc.items():
        print(k)
        for c,vals in sorted(v.items()):
            vals=sorted(vals); print("   %-28s median %.4g  (n=%d)"%(c,vals[len(vals)//2],len(vals)))
PY
