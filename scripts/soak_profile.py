"""Dev tool: host profile (cProfile) of the evaluation loop on a synthetic sequence: where a frame's wall time goes
besides the tracker's graph replays.  usage: soak_profile.py [n_frames] [Replica|TUM]"""
import cProfile, io, os, pathlib, pstats, sys, tempfile
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from gsplatloc_amd.data.dataset import Parser
from gsplatloc_amd.eval import evaluate_room
from gsplatloc_amd.synthetic import write_replica_sequence, write_tum_sequence

n = int(sys.argv[1]) if len(sys.argv) > 1 else 21
kind = sys.argv[2] if len(sys.argv) > 2 else "Replica"
root = pathlib.Path(tempfile.mkdtemp())
if kind == "TUM":
    write_tum_sequence(root, 640, 480, n)
    parser = Parser("TUM", "freiburg1_desk", normalize=True, input_folder=str(root))
else:
    write_replica_sequence(root, 640, 480, n)
    parser = Parser("Replica", "room0", normalize=True, input_folder=str(root))
evaluate_room(parser, num_iters=2000, max_frames=3, verbose=False)  # warm-up: library, allocator, first captures
pr = cProfile.Profile()
pr.enable()
res = evaluate_room(parser, num_iters=2000, max_frames=None, verbose=False)
pr.disable()
print({k: res[k] for k in ("frames", "mean_steps", "seconds", "frames_per_s")})
out = io.StringIO()
pstats.Stats(pr, stream=out).sort_stats("cumulative").print_stats(45)
print(out.getvalue()[:9000])
