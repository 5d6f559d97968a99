"""Build the library of another revision next to the working tree's, for A/B timing on ONE box in ONE gpurun call:
    python tools/ab_build.py <git-rev>            -> 6dof-.../libpedp_hip_<rev>.so
    PEDP_LIB=.../libpedp_hip_<rev>.so python tools/icp_only.py     (on the box)"""
import importlib.util, os, shutil, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "6dof-pose-estimation-and-defect-projection_amd")
rev = sys.argv[1]
tmp = tempfile.mkdtemp(prefix="pedp_ab_")
subprocess.run(f"git -C {ROOT} archive {rev} 6dof-pose-estimation-and-defect-projection_amd include | tar -x -C {tmp}", shell=True, check=True)
pkg = os.path.join(tmp, "6dof-pose-estimation-and-defect-projection_amd")
spec = importlib.util.spec_from_file_location("pedp_build_ab", os.path.join(pkg, "build.py"))
b = importlib.util.module_from_spec(spec)
spec.loader.exec_module(b)
out = os.path.join(PKG, f"libpedp_hip_{rev[:7]}.so")
b.build(force=True, verbose=False, out=os.path.join(pkg, "libpedp_hip.so"))
shutil.copy(os.path.join(pkg, "libpedp_hip.so"), out)
shutil.rmtree(tmp)
print(out)
