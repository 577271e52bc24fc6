import sys, time, torch, types
sys.path.insert(0, '/root/repo')
from customnerf_amd.sd import arch
from customnerf_amd.sd.guidance import StableDiffusion
opt = types.SimpleNamespace(cfg=7.5, lambda_sd=0.01, max_ratio=0.98, stage_time=False, iters=1000, log_loss_item=False)
t0 = time.time()
guide = StableDiffusion("cuda", "1.5", opt)
torch.cuda.synchronize(); print("build", time.time() - t0, "s; mem GB", torch.cuda.memory_allocated() / 1e9)
text = guide.synthetic_text_embeds()
def timeit(f, n=5, w=2):
    for _ in range(w): f()
    torch.cuda.synchronize(); t = time.time()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.time() - t) / n * 1e3
x = torch.zeros(2, 64, 64, 8, dtype=torch.float16, device="cuda"); x[..., :4].normal_()
tt = torch.full((2,), 500.0, device="cuda")
ctx = text.half()
with torch.no_grad():
    print("unet eager ms", timeit(lambda: guide.unet(x, tt, ctx)))
    print("unet graph ms", timeit(lambda: guide.unet.graphed(x, tt, ctx)))
img = torch.rand(1, 3, 128, 128, device="cuda", requires_grad=True)
def vae_f():
    with torch.no_grad(): return guide.encode_imgs(img, resize=(512, 512))
def vae_fb():
    lat = guide.encode_imgs(img, resize=(512, 512)); lat.sum().backward()
print("vae fwd ms", timeit(vae_f))
print("vae fwd+bwd ms", timeit(vae_fb))
def step():
    lat = guide.encode_imgs(img, resize=(512, 512))
    loss, _ = guide.train_step(lat, text)
    loss.backward()
print("sds step (vae f+b, unet) ms", timeit(step))
print("mem GB", torch.cuda.max_memory_allocated() / 1e9)
