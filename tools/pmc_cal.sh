# tools/pmc_cal.sh <tag>: FETCH_SIZE / WRITE_SIZE of kernels with known byte counts (tools/latbench/pmccal.hip) -> gpurun_out/<tag>/pmc_calibration.json
TAG=$1
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/$TAG; mkdir -p $O
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf $O/cal_$C
  timeout 120 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/cal_$C -o bench -- tools/latbench/pmccal > $O/cal_req.json 2> $O/cal_$C.err
  python tools/pmc_one.py $O/cal_$C $C $O/cal_$C.json
  rm -rf $O/cal_$C
done
python - <<PY
import json
o="$O"; req=json.loads(open(o+"/cal_req.json").read().strip().splitlines()[-1])
F=json.load(open(o+"/cal_FETCH_SIZE.json")); W=json.load(open(o+"/cal_WRITE_SIZE.json"))
res={"requested_bytes":req["requested"],"gather_count":req["gather_count"],"kernels":{}}
for k,b in req["requested"].items():
    f=[v[1] for n,v in F.items() if k in n]; w=[v[1] for n,v in W.items() if k in n]
    res["kernels"][k]={"requested_bytes":b,"FETCH_SIZE_bytes":f[0] if f else None,"WRITE_SIZE_bytes":w[0] if w else None,
        "fetch_over_requested":(f[0]/b if f else None),"write_over_requested":(w[0]/b if w else None)}
g=res["kernels"].get("cal_gather8_read")
if g and g["FETCH_SIZE_bytes"]: g["fetch_bytes_per_access"]=g["FETCH_SIZE_bytes"]/req["gather_count"]
s=res["kernels"].get("cal_scatter4_write")
if s and s["WRITE_SIZE_bytes"]: s["write_bytes_per_access"]=s["WRITE_SIZE_bytes"]/req["gather_count"]
json.dump(res,open(o+"/pmc_calibration.json","w"),indent=1); print(json.dumps(res,indent=1))
PY
