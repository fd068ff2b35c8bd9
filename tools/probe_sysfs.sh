#!/bin/bash
# what the GPU box exposes for in-process clock / power sampling (bench.py's roofline.sclk_mhz / power_w)
for d in /sys/class/drm/card*/device; do
  echo "== $d"; ls $d | tr '\n' ' ' | head -c 1500; echo
  for h in $d/hwmon/hwmon*; do echo "-- $h"; ls $h | tr '\n' ' '; echo; for f in power1_average power1_input power1_cap freq1_input freq1_label freq2_input freq2_label; do [ -r $h/$f ] && echo "$f: $(cat $h/$f 2>&1)"; done; done
  [ -r $d/pp_dpm_sclk ] && { echo "pp_dpm_sclk:"; cat $d/pp_dpm_sclk; }
  [ -r $d/gpu_metrics ] && echo "gpu_metrics: $(wc -c < $d/gpu_metrics) bytes"
done
python - <<'PY'
import time, subprocess
t=time.time(); out=subprocess.run(["rocm-smi","--showpower","--showclocks"],capture_output=True,text=True).stdout; print("rocm-smi", round(time.time()-t,3),"s"); print(out[:1500])
try:
    import amdsmi; print("amdsmi importable", amdsmi.__file__)
except Exception as e: print("amdsmi:", e)
PY
