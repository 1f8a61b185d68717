import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); import conftest  # noqa
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
from gsplat_attack.scenes import make_scene
from gsplat_attack.renderer import PipelineParams, render
from gsplat_attack import dist as gdist
import time
model, cams, _ = make_scene("nyc-1M", device=torch.device("cuda:0"), n_views=1)
gc = torch.randn(3,1080,1920, device="cuda")
# monkeypatch world size check: call the flat path directly
for i in range(3):
    model.zero_grad()
    render(cams[0], model, PipelineParams(skip_objects=True), torch.zeros(3, device="cuda"))["render"].backward(gc)
    grads=[getattr(model,n).grad for n in gdist.ATTACK_PARAMS]
    flat=gdist._flat_view_of(grads)
    torch.cuda.synchronize(); t=time.perf_counter()
    dist.all_reduce(flat)
    torch.cuda.synchronize(); print("allreduce(world=1) of", flat.numel()*4/1e6, "MB:", (time.perf_counter()-t)*1e3, "ms")
dist.barrier(); dist.destroy_process_group(); print("ok")
