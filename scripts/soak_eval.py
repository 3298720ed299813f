"""Dev tool: the evaluation CLI protocol on a longer synthetic sequence (default 40 frame pairs).
usage: soak_eval.py [n_frames] [Replica|TUM]   (TUM: patches of invalid depth in every frame -- piles when the camera
steps backwards)"""
import json, pathlib, sys, tempfile, time
sys.path.insert(0, ".")
from gsplatloc_amd.data.dataset import Parser
from gsplatloc_amd.eval import evaluate_room
from gsplatloc_amd.synthetic import write_replica_sequence, write_tum_sequence

n = int(sys.argv[1]) if len(sys.argv) > 1 else 41
kind = sys.argv[2] if len(sys.argv) > 2 else "Replica"
root = pathlib.Path(tempfile.mkdtemp())
if kind == "TUM":
    write_tum_sequence(root, 640, 480, n)
    parser = Parser("TUM", "freiburg1_desk", normalize=True, input_folder=str(root))
else:
    write_replica_sequence(root, 640, 480, n)
    parser = Parser("Replica", "room0", normalize=True, input_folder=str(root))
t = time.perf_counter()
res = evaluate_room(parser, num_iters=2000, max_frames=None, verbose=False)
res["dataset"] = kind
print(json.dumps(res))
