"""Build libpedp_hip.so and the diagnostic libpedp_hip_stamps.so (-DPEDP_ICP_STAMPS=1, for tools/icp_stamps.py): python tools/build_libs.py"""
import importlib.util, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "6dof-pose-estimation-and-defect-projection_amd")
spec = importlib.util.spec_from_file_location("pedp_build", os.path.join(PKG, "build.py"))
b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
b.build(force=True, verbose=False)
b.build(force=True, verbose=False, extra_flags=["-DPEDP_ICP_STAMPS=1"], out=os.path.join(PKG, "libpedp_hip_stamps.so"))
