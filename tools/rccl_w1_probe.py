"""isolating probe for tests/test_rccl_world1_gpu.py: which switch makes the world-1 RCCL step differ from the single-process one"""
import os, subprocess, sys, socket
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_rccl_world1_gpu import SCRIPT
def run(envs):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0", **envs)
    env.pop("WORLD_SIZE", None)
    p = subprocess.run([sys.executable, "-c", SCRIPT, ROOT, "/tmp/probe.pt"], env=env, capture_output=True, text=True)
    print(envs, p.stdout.strip().splitlines()[-1] if p.stdout.strip() else p.stderr[-500:], flush=True)
run({"PASSION_DDP_FORCE": "1"})
run({"PASSION_DDP_FORCE": "1"})
run({"PASSION_DDP_FORCE": "0"})
run({"PASSION_DDP_FORCE": "0", "PASSION_EARLY_STEP": "0"})
run({"PASSION_DDP_FORCE": "0", "PASSION_EARLY_STEP": "0", "PASSION_DIRECT_GRADS": "0"})
run({"PASSION_DDP_FORCE": "1", "PASSION_DIST_BACKEND": "gloo"})
