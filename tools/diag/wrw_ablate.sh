#!/bin/bash
# conv_hwrw_kernel<4,2,2>: shipped build, its ablations (make -C uaps_amd/csrc wrwabl) and the geometry-free probe, one box.
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/wrw_ablate.txt
: > $out
for v in "" ${WRW_VARIANTS:-_wrwe1 _wrwe2 _wrwe4 _wrwe8 _wrwe15}; do
  echo "== libuaps_hip$v.so" >> $out
  UAPS_HIP_LIB=$R/uaps_amd/lib/libuaps_hip$v.so timeout 300 python3 $R/tools/diag/wrw_ablate.py 2>&1 | grep -v amdgpu.ids >> $out
done
echo "== probe (tools/split_wrw_ceiling.hip)" >> $out
timeout 300 $R/tools/bin/split_wrw_ceiling 2>&1 | grep -E "^[0-9]|shipped blocking|64x64 per workgroup" | cut -c1-230 >> $out
cat $out
