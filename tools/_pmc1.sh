set -e
ROOT=$GRAFT_REPO_ROOT
mkdir -p $ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmcf -o pmc -- python3 $ROOT/tools/hessian_probe.py --defer 16 11008 > /tmp/pmcf.log 2>&1
python3 - <<'PY'
import csv, sys
csv.field_size_limit(1<<30)
rows=[r for r in csv.DictReader(open('/tmp/pmcf/pmc_counter_collection.csv')) if 'hessian16_big16' in r['Kernel_Name'] and r['Counter_Name']=='FETCH_SIZE']
for r in rows[-12:]:
    d=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3
    print(r['Kernel_Name'][:40], r['Grid_Size'], f"{d:8.1f} us  fetch {2*1024*float(r['Counter_Value'])/1e9:6.2f} GB")
PY
