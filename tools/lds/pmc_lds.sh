#!/bin/bash
# Run ON the GPU box: LDS bank-conflict counters of the two step kernels for one build of the module
# usage: tools/lds/pmc_lds.sh <label> [bench args...]     (DATUM_OCEAN_HIP_LIB selects the build)
label=$1; shift
export TMPDIR=/tmp
out=gpurun_out/r05/pmc_lds_$label
rm -rf $out; mkdir -p $out
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $out -- python3 bench.py --cpu-seconds 0 --no-frame --no-regime --no-check "$@" > $out/log.txt 2>&1
python3 - "$out" "$label" <<'PY'
import csv, glob, sys, collections, re
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        m = re.search(r"ocean_(rowpass|colpass)_kernel<(\d+), (true|false)", row.get("Kernel_Name", ""))
        if m:
            acc[f"{m.group(1)}<{m.group(2)}{',h16' if m.group(3) == 'true' else ''}>"][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, c in sorted(acc.items()):
    mean = {n: sum(v) / len(v) for n, v in c.items()}
    print(f"{sys.argv[2]:<12} {k:<18} launches {len(c['SQ_LDS_IDX_ACTIVE']):5d}  SQ_LDS_BANK_CONFLICT {mean['SQ_LDS_BANK_CONFLICT']:12.0f}  SQ_LDS_IDX_ACTIVE {mean['SQ_LDS_IDX_ACTIVE']:12.0f}  ratio {mean['SQ_LDS_BANK_CONFLICT'] / max(1.0, mean['SQ_LDS_IDX_ACTIVE']):6.1%}"
          f"  SQ_INSTS_LDS {mean['SQ_INSTS_LDS']:10.0f}  SQ_WAIT_INST_LDS {mean['SQ_WAIT_INST_LDS']:11.0f}  SQ_ACTIVE_INST_LDS {mean['SQ_ACTIVE_INST_LDS']:11.0f}  SQ_WAVE_CYCLES {mean['SQ_WAVE_CYCLES']:12.0f}")
PY
rm -rf $out
