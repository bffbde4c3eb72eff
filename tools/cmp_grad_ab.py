"""A/B of two library builds on the C3 update: runs tools/perf_update.py once with UAVPPO_LIB=tools/bin/libuavppo_base.so (a copy of the
library built from the commit to compare with) and once with the in-tree library, and compares the flat gradients bit for bit."""
import os, sys, subprocess, numpy as np
ROOT = "/root/repo" if os.path.isdir("/root/repo") else os.getcwd()
outs = {}
for tag, lib in (("base", "tools/bin/libuavppo_base.so"), ("new", "")):
    env = dict(os.environ)
    if lib: env["UAVPPO_LIB"] = lib
    subprocess.run([sys.executable, "tools/perf_update.py"], env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True)
    outs[tag] = np.load("gpurun_out/grad_x6.npy")
print("gradients of the C3 update, base vs new library: identical bits:", np.array_equal(outs["base"], outs["new"]), " max |diff|", np.abs(outs["base"] - outs["new"]).max())
