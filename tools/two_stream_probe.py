"""Probe: do two half-batch refiner chains on two HIP streams pack the machine better than one full-batch chain?
(Conv launches leave CUs idle in their last, partially filled round; hypotheses are independent.)  Perf only: the
K-slice workspaces are process-wide, so run with HP_CONV_NO_SPLITK=1 when the poses matter."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from happypose_amd.models import create_pose_model_cosypose

dev = torch.device("cuda:0")
ds, renderer, scene, weights, model = bench.build_world(dev, "resnet34", 0, "C2", "f32")
store = renderer.store
B = len(scene["TCO_hyp"])
images = torch.as_tensor(scene["images"], device=dev); K = torch.as_tensor(scene["K"], device=dev)
TCO0 = torch.as_tensor(scene["TCO_hyp"], device=dev)
labels = [store.labels[i] for i in scene["hyp_obj_ids"]]
im_ids = torch.zeros(B, dtype=torch.int32, device=dev)
NL = int(os.environ.get("LANES", 2))
from happypose_amd.renderer import BatchRenderer
halves = [create_pose_model_cosypose(dict(backbone_str="resnet34"), BatchRenderer(ds, device=dev), state_dict=weights,
                                     max_batch=(B + NL - 1) // NL, precision="f32") for _ in range(NL)]
streams = [torch.cuda.Stream(device=dev) for _ in range(NL)]


def full():
    return model.forward(images, K, labels, TCO0, n_iterations=5, im_ids=im_ids)["iteration=5"].TCO_output


def split():
    outs = []
    cur = torch.cuda.current_stream(dev)
    for h in range(NL):
        sl = slice(h * B // NL, (h + 1) * B // NL)
        streams[h].wait_stream(cur)
        with torch.cuda.stream(streams[h]):
            outs.append(halves[h].forward(images, K, labels[sl], TCO0[sl], n_iterations=5, im_ids=im_ids[sl])["iteration=5"].TCO_output)
    for s in streams:
        cur.wait_stream(s)
    return torch.cat(outs)


for name, fn in (("full", full), ("two-stream", split), ("full", full), ("two-stream", split)):
    for _ in range(3):
        p = fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        p = fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    print(f"{name:10s} {1e3 * dt:7.2f} ms/step  {B / dt:8.1f} poses/s")
print("max |full - two-stream| =", float((full() - split()).abs().max()))
