#!/bin/bash
# Development helper (GPU box): what clock / power / temperature telemetry can an ordinary user read here?
for d in /sys/class/drm/card*/device; do
  echo "== $d"; cat $d/unique_id 2>/dev/null
  for f in pp_dpm_sclk pp_dpm_mclk gpu_busy_percent current_link_speed pp_power_profile_mode; do echo "-- $f"; head -c 600 $d/$f 2>&1 | head -12; done
  for h in $d/hwmon/hwmon*; do echo "-- $h"; ls $h | tr '\n' ' '; echo; for f in power1_average power1_input power1_cap freq1_input freq2_input temp1_input temp2_input temp3_input; do [ -e $h/$f ] && echo "$f = $(cat $h/$f 2>&1)"; done; done
done 2>&1 | head -120
python3 - <<'PY'
import sys
for p in ("/opt/rocm/share/amd_smi", "/opt/rocm/libexec/rocm_smi"):
    sys.path.insert(0, p)
try:
    import amdsmi
    amdsmi.amdsmi_init()
    hs = amdsmi.amdsmi_get_processor_handles()
    print("amdsmi ok:", len(hs), "handles")
    h = hs[0]
    for fn in ("amdsmi_get_clock_info", "amdsmi_get_power_info", "amdsmi_get_gpu_metrics_info"):
        try:
            r = getattr(amdsmi, fn)(h, amdsmi.AmdSmiClkType.GFX) if fn == "amdsmi_get_clock_info" else getattr(amdsmi, fn)(h)
            print(fn, {k: r[k] for k in list(r)[:40]} if isinstance(r, dict) else r)
        except Exception as e:
            print(fn, "->", repr(e))
except Exception as e:
    print("amdsmi import/init failed:", repr(e))
PY
which rocm-smi amd-smi; timeout 20 rocm-smi --showclocks --showpower --showtemp 2>&1 | head -40
