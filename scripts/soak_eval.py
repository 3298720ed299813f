"""Dev tool: the evaluation CLI protocol on a longer synthetic Replica-format sequence (default 40 frame pairs)."""
import json, pathlib, sys, tempfile, time
sys.path.insert(0, ".")
from gsplatloc_amd.data.dataset import Parser
from gsplatloc_amd.eval import evaluate_room
from gsplatloc_amd.synthetic import write_replica_sequence

n = int(sys.argv[1]) if len(sys.argv) > 1 else 41
root = pathlib.Path(tempfile.mkdtemp())
write_replica_sequence(root, 640, 480, n)
parser = Parser("Replica", "room0", normalize=True, input_folder=str(root))
t = time.perf_counter()
res = evaluate_room(parser, num_iters=2000, max_frames=None, verbose=False)
print(json.dumps(res))
