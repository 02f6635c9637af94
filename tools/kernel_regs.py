"""Dev tool (no GPU needed): registers, spills and scratch of the library's kernels, from the code objects' metadata.
    python3 tools/kernel_regs.py [regex]      e.g.  python3 tools/kernel_regs.py point_pass"""
import os, re, shutil, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
pat = re.compile(sys.argv[1] if len(sys.argv) > 1 else ".")
KEYS = "name|vgpr_count|agpr_count|sgpr_count|vgpr_spill_count|sgpr_spill_count|private_segment_fixed_size|group_segment_fixed_size"
with tempfile.TemporaryDirectory() as tmp:
    shutil.copy(os.environ.get("DZO_LIB_PATH", os.path.join(ROOT, "dzoptimization.jl_amd", "libdzo_hip.so")), os.path.join(tmp, "lib.so"))
    subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", "lib.so"], cwd=tmp, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    rows = {}
    for f in sorted(os.listdir(tmp)):
        if "gfx950" not in f:
            continue
        notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", f], cwd=tmp, check=True, capture_output=True, text=True).stdout
        # one YAML map per kernel: entries start with "  - .xxx:"; collect the keys of each map
        cur = None
        for line in notes.splitlines():
            if re.match(r"\s+- \.", line):
                if cur and "name" in cur and "vgpr_count" in cur:
                    rows[cur["name"]] = cur
                cur = {}
            m = re.match(r"\s+(?:- )?\.(" + KEYS + r"):\s+(\S+)", line)
            if m and cur is not None:
                cur[m.group(1)] = m.group(2)
        if cur and "name" in cur and "vgpr_count" in cur:
            rows[cur["name"]] = cur
    for n, r in sorted(rows.items()):
        if not pat.search(n):
            continue
        dem = subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip()
        dem = re.sub(r"\(dzo::\w+<\w+>\)", "", dem).replace("void dzo::", "")
        print(f"{dem:70s} vgpr {r.get('vgpr_count','?'):>4} agpr {r.get('agpr_count','?'):>3} sgpr {r.get('sgpr_count','?'):>3} "
              f"vspill {r.get('vgpr_spill_count','0'):>3} sspill {r.get('sgpr_spill_count','0'):>3} scratch {r.get('private_segment_fixed_size','0'):>4} lds {r.get('group_segment_fixed_size','0')}")
