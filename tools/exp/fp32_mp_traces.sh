for sel in "1152,8" "672,16" "240,32"; do
  for e in "HEP_MBF_MP_RES=1" "HEP_MBF_MP_RES=0" "HEP_MBF_MP=0"; do echo "== mbf $sel fp32 $e"; env $e HEP_MBF_TRACE_SEL=$sel python tools/trace_mbf.py 16 300 fp32 2>&1 | grep -v amdgpu.ids | grep -v "start hist"; done
done
