"""Clock / power telemetry beside a measurement (review r5 item 6): a side thread samples the GPU's graphics clock, socket power, power
cap and hotspot temperature every `period_s` while a timed region runs, so that "power-limited ~1.4-1.7 GHz" and "boxes differ by 10 %"
are numbers next to the numbers they explain, not inferences from cycle counters.

    with Telemetry(device_index=0, period_s=0.05) as tm:
        ... timed region ...
    tm.summary()   # {"source": ..., "samples": n, "sclk_mhz": {"mean", "min", "max"}, "power_w": {...}, "power_cap_w": ..., ...}

Sources, first one that works: the `amdsmi` Python bindings of the ROCm image (gpu_metrics: per-XCD current gfx clocks, socket power,
throttle status), then sysfs hwmon (freq1_input, power1_average / power1_input, power1_cap).  No source: summary() says so - the bench
line then carries `telemetry: {"source": null}` instead of a guess.  Sampling costs one SMU query per period on a host thread; it does
not touch the HIP streams.  Used by bench.py (the timed region) and tools/gemm_shapes.py / tools/clock_probe.py (per problem class)."""
import glob
import os
import threading
import time


def _num(x):
    """amdsmi reports 'N/A' strings, ints or lists: -> float or None"""
    try:
        if isinstance(x, (list, tuple)):
            vals = [float(v) for v in x if isinstance(v, (int, float)) and 0 < float(v) < 60000]
            return sum(vals) / len(vals) if vals else None
        v = float(x)
        return v if v == v and 0 <= v < 1e7 else None
    except (TypeError, ValueError):
        return None


class _AmdSmi:
    name = "amdsmi"

    def __init__(self, index):
        import amdsmi
        self.smi = amdsmi
        amdsmi.amdsmi_init()
        hs = amdsmi.amdsmi_get_processor_handles()
        if not hs:
            raise RuntimeError("amdsmi: no processors")
        self.h = hs[min(index, len(hs) - 1)]
        self.cap = None
        try:
            pi = amdsmi.amdsmi_get_power_info(self.h)
            self.cap = _num(pi.get("power_limit"))
            if self.cap and self.cap > 1e5:        # some versions report microwatts
                self.cap /= 1e6
        except Exception:      # noqa: BLE001
            pass
        self.sample()          # raises if nothing usable comes back

    def sample(self):
        m = self.smi.amdsmi_get_gpu_metrics_info(self.h)
        xcd = m.get("current_gfxclks")
        per_xcd = [float(v) for v in xcd if isinstance(v, (int, float)) and 0 < float(v) < 60000] if isinstance(xcd, (list, tuple)) else []
        sclk = (sum(per_xcd) / len(per_xcd)) if per_xcd else _num(m.get("current_gfxclk"))
        if sclk is None:
            sclk = _num(m.get("average_gfxclk_frequency"))
        power = _num(m.get("current_socket_power"))
        if power is None:
            power = _num(m.get("average_socket_power"))
        if sclk is None and power is None:
            raise RuntimeError("amdsmi gpu_metrics carries neither a gfx clock nor a socket power")
        return dict(sclk=sclk, sclk_min_xcd=min(per_xcd) if per_xcd else None, power=power, temp=_num(m.get("temperature_hotspot")),
                    mclk=_num(m.get("current_uclk")), throttle=m.get("throttle_status") if isinstance(m.get("throttle_status"), int) else None,
                    activity=_num(m.get("average_gfx_activity")))


class _Sysfs:
    name = "sysfs hwmon"

    def __init__(self, index):
        cards = sorted(glob.glob("/sys/class/drm/card[0-9]*/device/hwmon/hwmon*"))
        cards = [c for c in cards if os.path.exists(os.path.join(c, "freq1_input")) or os.path.exists(os.path.join(c, "power1_average"))
                 or os.path.exists(os.path.join(c, "power1_input"))]
        if not cards:
            raise RuntimeError("no amdgpu hwmon directory")
        self.d = cards[min(index, len(cards) - 1)]
        self.cap = self._read("power1_cap", 1e-6)
        self.sample()

    def _read(self, name, scale):
        try:
            with open(os.path.join(self.d, name)) as f:
                return float(f.read().strip()) * scale
        except (OSError, ValueError):
            return None

    def sample(self):
        sclk = self._read("freq1_input", 1e-6)
        power = self._read("power1_average", 1e-6)
        if power is None:
            power = self._read("power1_input", 1e-6)
        if sclk is None and power is None:
            raise RuntimeError("hwmon carries neither freq1_input nor power1_*")
        return dict(sclk=sclk, sclk_min_xcd=None, power=power, temp=self._read("temp2_input", 1e-3), mclk=self._read("freq2_input", 1e-6), throttle=None,
                    activity=None)


def open_source(index=0):
    errors = []
    for cls in (_AmdSmi, _Sysfs):
        try:
            return cls(index), errors
        except Exception as e:      # noqa: BLE001 - a missing source is an answer, not a failure
            errors.append(f"{cls.name}: {type(e).__name__}: {e}")
    return None, errors


class Telemetry:
    def __init__(self, device_index=0, period_s=0.05):
        self.period = period_s
        self.src, self.errors = open_source(device_index)
        self.rows = []
        self._stop = threading.Event()
        self._th = None

    def _loop(self):
        while not self._stop.is_set():
            t = time.perf_counter()
            try:
                self.rows.append((t, self.src.sample()))
            except Exception as e:      # noqa: BLE001
                self.errors.append(f"sample: {type(e).__name__}: {e}")
                if len(self.errors) > 20:
                    return
            self._stop.wait(max(0.0, self.period - (time.perf_counter() - t)))

    def __enter__(self):
        if self.src is not None:
            self._th = threading.Thread(target=self._loop, name="vcx-telemetry", daemon=True)
            self._th.start()
        return self

    def __exit__(self, *exc):
        self._stop.set()
        if self._th is not None:
            self._th.join(timeout=2.0)
        return False

    def summary(self, t0=None, t1=None):
        """Statistics of the samples taken in [t0, t1] (perf_counter times; default: all)."""
        if self.src is None:
            return {"source": None, "errors": self.errors[:4]}
        rows = [r for (t, r) in self.rows if (t0 is None or t >= t0) and (t1 is None or t <= t1)]

        def stat(key):
            v = [r[key] for r in rows if r.get(key) is not None]
            if not v:
                return None
            v.sort()
            return {"mean": round(sum(v) / len(v), 1), "min": round(v[0], 1), "p10": round(v[len(v) // 10], 1), "max": round(v[-1], 1)}
        thr = [r["throttle"] for r in rows if r.get("throttle") is not None]
        out = {"source": self.src.name, "period_s": self.period, "samples": len(rows), "sclk_mhz": stat("sclk"), "sclk_min_xcd_mhz": stat("sclk_min_xcd"),
               "power_w": stat("power"), "power_cap_w": self.src.cap, "temp_hotspot_c": stat("temp"), "mclk_mhz": stat("mclk"),
               "gfx_activity_pct": stat("activity")}
        if thr:
            out["throttle_status_or"] = 0
            for t in thr:
                out["throttle_status_or"] |= int(t)
        if self.errors:
            out["errors"] = self.errors[:4]
        return out


if __name__ == "__main__":      # python tools/telemetry.py: what this box offers, one second of idle samples
    import json
    tm = Telemetry(period_s=0.05)
    with tm:
        time.sleep(1.0)
    print(json.dumps(tm.summary(), indent=1))
    if isinstance(tm.src, _AmdSmi):
        m = tm.src.smi.amdsmi_get_gpu_metrics_info(tm.src.h)
        print({k: v for k, v in m.items() if "clk" in k or "power" in k or "throttle" in k or "temp" in k or "activity" in k})
        try:
            print(tm.src.smi.amdsmi_get_power_info(tm.src.h))
            print(tm.src.smi.amdsmi_get_clock_info(tm.src.h, tm.src.smi.AmdSmiClkType.GFX))
        except Exception as e:      # noqa: BLE001
            print("power/clock info:", e)
