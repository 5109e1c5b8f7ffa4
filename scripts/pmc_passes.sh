#!/bin/bash
# PMC passes over `bench.py` (one rocprofv3 run per pass; counters never combined with tracing).
# usage: scripts/pmc_passes.sh <outdir>            (profiles `python3 bench.py --steps 1 --warmup 0 --cpu-seconds 0`)
#        PMC_CMD="scripts/run_config.py c5 --steps 1" scripts/pmc_passes.sh <outdir>
set -u
out=$1; shift
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
pass() {
  name=$1; shift
  timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d "$out/$name" -- python3 ${PMC_CMD:-bench.py --steps 1 --warmup 0 --cpu-seconds 0} > "$out/$name.log" 2>&1 || { echo "pass $name failed"; tail -3 "$out/$name.log"; return 1; }
}
pass sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE &&
pass sq2 SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_INT32 &&
pass sq3 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT64 SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM &&
pass sq4 SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_IOPS &&
pass tcc1 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum &&
pass tcc2 FETCH_SIZE &&
pass tcc3 WRITE_SIZE TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum
python3 - "$out" <<'PY'
import csv, glob, re, sys, collections
out = sys.argv[1]
tot = collections.OrderedDict()
for f in sorted(glob.glob(out + '/*/*/*counter_collection.csv')):
    for row in csv.DictReader(open(f)):
        if 'render_kernel' not in row['Kernel_Name']: continue
        if re.search(r'render_kernel_ws<\w+, \d+, [123](, \w+)*>', row['Kernel_Name']): continue  # diagnostic builds (bench.py's segment count launch)
        tot[row['Counter_Name']] = tot.get(row['Counter_Name'], 0.0) + float(row['Counter_Value'])
with open(out + '/summary.txt', 'w') as fh:
    for k, v in tot.items():
        line = '%-28s %.6g' % (k, v); print(line); fh.write(line + '\n')
PY
