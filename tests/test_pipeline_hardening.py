"""What the first N-GPU run must survive: registrations that fail, and callers whose heap hands
a file's memory out again — for another file, or for pixels — while the pipeline keeps (or has just dropped) a
registration of it.  Each in a process of its own: a GPU memory fault ends the process it happens in."""
import os
import subprocess
import sys
import textwrap

import pytest

from conftest import ROOT


def _run(prog, env=None, timeout=600):
    e = dict(os.environ)
    e.update(env or {})
    r = subprocess.run([sys.executable, "-c", textwrap.dedent(prog) % {"root": ROOT}], capture_output=True, text=True,
                       timeout=timeout, env=e)
    assert r.returncode == 0 and "fine" in r.stdout, (r.returncode, r.stdout[-400:], r.stderr[-1200:])
    return r.stdout


@pytest.mark.gpu
def test_pipeline_when_no_buffer_can_be_registered(gpu):
    """hipHostRegister fails (RLIMIT_MEMLOCK of 8 MB for a process that may not override it; for root, who may, the
    tuning build's JGA_PIPE_REGISTER_FAIL makes every registration fail the same way): the files are copied into the
    groups' pinned blobs instead, nothing is registered, and every image decodes to the oracle's pixels — long run
    and short run, clean-up on the device."""
    out = _run("""
        import resource, sys
        resource.setrlimit(resource.RLIMIT_MEMLOCK, (8 << 20, 8 << 20))
        sys.path.insert(0, %(root)r)
        import numpy as np
        from jpeg_gpu_amd import abi, lib, synth
        import oracle
        orc = oracle.Oracle()
        files = [synth.synthetic_jpeg(1280, 720, "420", quality=90, seed=70 + i) for i in range(6)]
        want = [orc.decode_rgb(f)[1].reshape(-1) for f in files]
        arrs = [np.frombuffer(f, np.uint8).copy() for f in files]
        for n in (6, 60):                                        # one group per lane / several
            for mb in (0, 64):                                   # run-scoped registrations / the persistent cache
                pl = lib.Pipeline(device=0, nthreads=4, out=abi.JPEG_DECODE_RGB, copy_back=True, transport=2, batch=1,
                                  depth=3, unstuff=2, input_cache_mb=mb)
                outs = [np.zeros(want[0].size, np.uint8) for _ in range(n)]
                jobs = lib.Pipeline.make_jobs([arrs[i %% 6] for i in range(n)], host_outs=outs)
                assert pl.run_jobs(jobs) == 0 and all(j.status == 0 for j in jobs)
                c = pl.counters()
                pl.close()
                assert all(np.array_equal(outs[i], want[i %% 6]) for i in range(n))
                print(n, mb, c["registered"], c["jobs_in_place"], c["jobs_copied"])
                assert c["registered"] == 0 and c["jobs_in_place"] == 0 and c["jobs_copied"] == n, c
        print("fine")
    """, env={"JGA_LIB_PATH": os.path.join(ROOT, "jpeg_gpu_amd", "libjpeg_gpu_amd_tuning.so"),
              "JGA_PIPE_REGISTER_FAIL": "1"})
    assert "fine" in out


@pytest.mark.gpu
@pytest.mark.parametrize("cache_mb", [0, 64])
def test_pipeline_when_the_callers_heap_recycles_files_and_pixels(gpu, cache_mb):
    """The plugin's heap-recycling stress (tests/test_harness.py) through jga_pipeline_run: the caller mallocs a
    file, decodes it with the pixels copied back into a freshly malloc'd PAGEABLE buffer, frees both, and the heap
    (M_MMAP_THRESHOLD raised: everything comes from it) hands the same memory out again — a 6 MB file's for the
    next frame's 6 MB of pixels, a frame's for the next file.  With the default registrations (undone before the run
    returns) nothing is asked of the caller; with the persistent cache (input_cache_mb = 64) the caller forgets the
    buffer before it frees it, as the header says.  Forty rounds, every pixel against the oracle, no fault."""
    _run("""
        import ctypes as C, gc, sys
        libc = C.CDLL("libc.so.6")
        libc.mallopt(-3, 256 << 20)           # M_MMAP_THRESHOLD: the heap serves everything under 256 MB
        sys.path.insert(0, %%(root)r)
        import numpy as np
        from jpeg_gpu_amd import abi, lib, synth
        import oracle
        orc = oracle.Oracle()
        big = synth.synthetic_jpeg(3840, 2160, "444", quality=90, seed=7)       # a 6 MB file ...
        frame = synth.synthetic_jpeg(1920, 1080, "420", quality=90, seed=8)     # ... and a frame of 6 MB of RGB
        small = synth.synthetic_jpeg(1280, 720, "420", quality=90, seed=9)
        want = {id(f): orc.decode_rgb(f)[1].reshape(-1) for f in (big, frame, small)}
        cache_mb = %d
        pl = lib.Pipeline(device=0, nthreads=4, out=abi.JPEG_DECODE_RGB, copy_back=True, transport=2, batch=2, depth=3,
                          unstuff=2, input_cache_mb=cache_mb)
        for rep in range(40):
            order = [(big, frame, small), (frame, small, big), (small, big, frame)][rep %%%% 3]
            for f in order:
                buf = np.frombuffer(f, np.uint8).copy()                         # malloc
                outs = [np.empty(want[id(f)].size, np.uint8) for _ in range(2)] # malloc (pageable destination)
                jobs = lib.Pipeline.make_jobs([buf, buf], host_outs=outs)
                assert pl.run_jobs(jobs) == 0
                assert np.array_equal(outs[0], want[id(f)]) and np.array_equal(outs[1], want[id(f)]), rep
                if cache_mb > 0:
                    pl.forget_input(buf)                                        # (the persistent cache's contract)
                del jobs, buf, outs
                gc.collect()                                                    # free: the heap has it back
        c = pl.counters()
        pl.close()
        print(c)
        assert c["registered"] >= 40                                            # (the big files were read where they lay)
        print("fine")
    """ % cache_mb)
