#!/usr/bin/env python3
"""Kernel descriptors (registers, spills, scratch, LDS) of the worker / FM kernels as built: read from the gfx950 code objects inside the
   .hip_fatbin sections of bowtie2_amd/csrc/*.o.      tools/kernel_descriptors.py [name-substring ...]"""
import glob, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin/"
pats = sys.argv[1:] or ["k_align", "k_exact", "k_one_mm", "k_seed", "k_extend", "k_dp_fill"]
with tempfile.TemporaryDirectory() as td:
    for obj in sorted(glob.glob(os.path.join(ROOT, "bowtie2_amd", "csrc", "*.o"))):
        fb, co = os.path.join(td, "x.fatbin"), os.path.join(td, "x.co")
        if subprocess.run(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", obj, fb]).returncode or not os.path.getsize(fb): continue
        if subprocess.run([LLVM + "clang-offload-bundler", "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--input=" + fb, "--output=" + co],
                          capture_output=True).returncode: continue
        txt = subprocess.run([LLVM + "llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
        for blk in txt.split("  - .agpr_count:")[1:]:
            f = dict(re.findall(r"\.(\w+):\s+'?([^\n']+)'?", ".agpr_count:" + blk))
            name = f.get("name", "")
            if not any(p in name for p in pats): continue
            dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip().split("(")[0]
            print("%-22s %-58s vgpr %3s sgpr %3s  vspill %3s sspill %3s  scratch %5s B  lds %5s B" % (os.path.basename(obj), dem[-58:], f.get("vgpr_count"), f.get("sgpr_count"),
                  f.get("vgpr_spill_count"), f.get("sgpr_spill_count"), f.get("private_segment_fixed_size"), f.get("group_segment_fixed_size")))
