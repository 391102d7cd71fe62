# per-wave phase stamps of the fp32 fronts / project GEMMs (profiling build)
for sel in "1152,8" "672,16" "240,32" "480,16"; do
  for p in fp32 bf16; do echo "== mbf $sel $p"; HEP_MBF_TRACE_SEL=$sel python tools/trace_mbf.py 16 300 $p 2>&1 | grep -v amdgpu.ids; done
done
for sel in "1152,192" "672,112" "480,80"; do
  for p in fp32 bf16; do echo "== pw $sel $p"; HEP_PW_TRACE_SEL=$sel python tools/trace_pw.py 16 300 $p 2>&1 | grep -v amdgpu.ids; done
done
