#!/bin/bash
# What can an ordinary user read about clock and socket power on the GPU box?  (round 6: bench.py samples them.)
out=gpurun_out/power_probe; mkdir -p $out
{
echo "== whoami"; id
echo "== drm cards"; ls -d /sys/class/drm/card*/device 2>&1
for d in /sys/class/drm/card*/device; do
  echo "-- $d"; ls $d 2>/dev/null | tr '\n' ' '; echo
  for f in pp_dpm_sclk pp_dpm_mclk gpu_busy_percent current_link_speed; do [ -r $d/$f ] && { echo "[$f]"; cat $d/$f; }; done
  for h in $d/hwmon/hwmon*; do echo "hwmon $h"; ls $h | tr '\n' ' '; echo; for f in power1_average power1_input power1_cap power1_cap_max freq1_input freq2_input temp1_input name; do [ -r $h/$f ] && echo "$f = $(cat $h/$f 2>&1)"; done; done
done
echo "== which"; which rocm-smi amd-smi rocminfo
echo "== rocm-smi"; timeout 30 rocm-smi --showpower --showclocks --showmaxpower 2>&1 | tail -40
echo "== amd-smi metric"; timeout 30 amd-smi metric -p -c 2>&1 | tail -60
echo "== amd-smi static limit"; timeout 30 amd-smi static -l 2>&1 | tail -30
echo "== timing of one amd-smi call"; ( time timeout 30 amd-smi metric -p -c --json >/dev/null 2>&1 ) 2>&1 | tail -4
echo "== python amdsmi?"; python3 -c "import amdsmi; print(amdsmi.__file__)" 2>&1 | tail -1
PYTHONPATH=/opt/rocm/share/amd_smi python3 -c "
import amdsmi,time
amdsmi.amdsmi_init()
hs=amdsmi.amdsmi_get_processor_handles(); print(len(hs))
h=hs[0]
t=time.time()
for i in range(20):
    p=amdsmi.amdsmi_get_power_info(h); c=amdsmi.amdsmi_get_clock_info(h, amdsmi.AmdSmiClkType.GFX)
print((time.time()-t)/20, p, c)
try: print(amdsmi.amdsmi_get_gpu_metrics_info(h))
except Exception as e: print('metrics', e)
" 2>&1 | tail -30
} > $out/probe.txt 2>&1
tail -5 $out/probe.txt
