cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/fx_pmc
rm -rf $O && mkdir -p $O
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC" "SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_FLAT" FETCH_SIZE WRITE_SIZE; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/g$i -- python3 tools/gpu_fusedx_phases.py > $O/g$i.log 2>&1
done
python3 - <<'P'
import csv, glob, collections
tot=collections.defaultdict(list)
for f in glob.glob('gpurun_out/fx_pmc/g*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'k_vae_fusedx' in r['Kernel_Name']:
            tot[r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in sorted(tot.items()):
    print(f"{k:32s} {sum(v)/len(v):16.0f}  (n={len(v)})")
P
