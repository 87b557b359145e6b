import sys, time, torch
sys.path.insert(0, "/root/repo")
import ssg_amd
from oracle import embed_oracle
sd = ssg_amd.synthetic_state_dict(seed=1)
imgs = torch.randn(32, 3, 256, 128, generator=torch.Generator().manual_seed(1))
for th in (256, 128, 64, 32):
    torch.set_num_threads(th)
    t0 = time.time(); embed_oracle.embed_with_flip(sd, imgs[:8], 1); t1 = time.time()
    embed_oracle.embed_with_flip(sd, imgs, 1); t2 = time.time()
    print("threads %d: warm-up (8 img) %.2f s, 32 images %.2f s -> %.1f img/s" % (th, t1 - t0, t2 - t1, 32 / (t2 - t1)), flush=True)
