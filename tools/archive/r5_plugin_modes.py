"""Which history puts the plugin's decode_image(RGB) of a 4K frame into its slow mode (+0.4 ms)?  Each sequence of steps
runs in a fresh process; the last step's plugin time is printed.  Steps: L<f> pipeline latency, P<f> plugin, D<f> device
only; <f> = 1080 / 4k / 444.  Usage: python tools/archive/r5_plugin_modes.py [sequence ...]   (no argument: the sweep)"""
import sys, os, subprocess, time
sys.path.insert(0, "."); sys.path.insert(0, "tools")

FILES = {"1080": (1920, 1080, "420"), "4k": (3840, 2160, "420"), "444": (3840, 2160, "444")}


def run(seq):
    from jpeg_gpu_amd import abi, lib, synth
    import configs_bench as cb
    out = []
    for step in seq.split(","):
        kind, name = step[0], step[1:]
        w, h, s = FILES[name]
        f = synth.synthetic_jpeg(w, h, s, quality=90, seed=1234)
        if kind == "L":
            out.append("%s %.3f" % (step, min(cb._pipeline_latency(lib, abi, f, 8, reps=20) for _ in range(3)) * 1e3))
        elif kind == "P":
            out.append("%s %.3f" % (step, cb._plugin(lib, abi, f, 20)["ms_per_frame"]))
        elif kind == "D":
            out.append("%s %.3f" % (step, cb._device_only(lib, [f], 1, 8)["ms"]))
    print("%-40s %s" % (seq, " | ".join(out)), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        run(sys.argv[1])
    else:
        for seq in ("P4k", "P4k,P4k", "L4k,P4k", "D4k,P4k", "L1080,P1080,D1080,L4k,P4k", "L1080,P1080,D1080,L4k,P4k,D4k,L444,P444",
                    "P1080,P4k,P444", "P444", "P4k,P444,P4k,P444", "L1080,P4k", "D1080,P4k", "L444,P444", "D444,P444"):
            for env in ({}, {"JGA_HUFF_NO_WIDE": "1"}):
                e = dict(os.environ); e.update(env)
                r = subprocess.run([sys.executable, __file__, seq], env=e, capture_output=True, text=True, timeout=300)
                print(("nowide " if env else "wide   ") + (r.stdout.strip() or r.stderr.strip()[-300:]), flush=True)
