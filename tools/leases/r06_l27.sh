#!/bin/bash
# round 6, lease 27: tile order of conv_gemm_x6 (groups of row panels, m fastest) against the n-fastest order of rounds 1-5 on the
# GEMM records of the batch-200 forward: time per record (whole chip / 128-CU partition), forced group heights, and FETCH / WRITE
# counter passes of both orders
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r06aa; mkdir -p $O
X=./audioeditingcode_amd/x6_bench
timeout 120 $X 1 cases > $O/cases.log 2>&1; echo "feature cases rc=$?"; tail -2 $O/cases.log
for v in 0 2 3 4; do
  timeout 300 $X 5 replay profiles/unet_b200_share2_gemm_ops.txt order gm=$v > $O/order_chip_gm$v.jsonl 2> $O/order_chip_gm$v.err; echo "chip gm=$v rc=$? $(tail -1 $O/order_chip_gm$v.jsonl | cut -c1-300)"
done
for v in 0 2 3; do
  timeout 300 $X 5 replay profiles/unet_b200_cus128_share2_gemm_ops.txt cus=128 order gm=$v > $O/order_cus128_gm$v.jsonl 2> $O/order_cus128_gm$v.err; echo "cus128 gm=$v rc=$? $(tail -1 $O/order_cus128_gm$v.jsonl | cut -c1-300)"
done
export TMPDIR=/tmp; R=$PWD; cd /tmp
for var in old new; do
  fl=0; [ $var = old ] && fl=1024
  timeout 300 rocprofv3 --pmc FETCH_SIZE WRITE_SIZE -d $R/$O/pmc_$var -o pmc --output-format csv -- $R/audioeditingcode_amd/x6_bench 1 replay $R/profiles/unet_b200_share2_gemm_ops.txt x6only allflags=$fl > $R/$O/pmc_$var.log 2>&1; echo "pmc $var rc=$?"
done
cd $R
python - <<'PY'
import csv,glob,collections
for var in ('old','new'):
    tot=collections.defaultdict(lambda:[0,0.0,0.0])
    for f in glob.glob(f'gpurun_out/r06aa/pmc_{var}/**/*counter_collection.csv', recursive=True):
        for row in csv.DictReader(open(f)):
            k=row['Kernel_Name'].split('(')[0][:60]
            t=tot[k]
            if row['Counter_Name']=='FETCH_SIZE': t[1]+=float(row['Counter_Value']); t[0]+=1
            if row['Counter_Name']=='WRITE_SIZE': t[2]+=float(row['Counter_Value'])
    for k,v in sorted(tot.items(), key=lambda kv:-kv[1][1])[:8]:
        print(var, k, v[0], 'fetch KBx2=%.1f GB'%(v[1]*2/1e6), 'write %.1f GB'%(v[2]/1e6))
PY
rm -rf $O/pmc_old/*/*agent_info* 2>/dev/null; du -sh $O
