"""k_extend_hits extends a one-row seed hit by comparing the read with the reference text (fm_extend_hit_text, eight characters per memory round
trip) instead of walking LF as SwDriver::extend does (aligner_sw_driver.cpp:299-484).  The host twin built with BT2G_CHECK_EXTEND_TEXT runs
both forms on every one-row hit it extends and aborts on the first disagreement."""
import os
import subprocess

from bt2test import HOSTSIM_CLASS_FLAGS
from test_work_counters import workload

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HS = os.path.join(ROOT, "tests", "hostsim")


def test_text_extension_equals_lf_walk(tmp_path):
    exe = str(tmp_path / "hostsim_chk")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-w"] + HOSTSIM_CLASS_FLAGS + ["-DBT2G_CHECK_EXTEND_TEXT", "-I" + os.path.join(ROOT, "include"), "-o", exe,
                           os.path.join(HS, "hostsim.cpp"), os.path.join(ROOT, "bowtie2_amd", "csrc", "bt2g_index.cpp"), "-lz", "-lpthread"])
    for large in (False, True):
        base, fq = workload(large)
        for args in ([], ["--local"], ["--very-sensitive"]):
            p = subprocess.run([exe] + args + ["-x", base, "-U", fq, "-S", os.devnull], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
            assert p.returncode == 0 and "extend mismatch" not in p.stderr, (large, args, p.stderr[-300:])
