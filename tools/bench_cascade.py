"""BASELINE config C5, per-GPU share: cascade shared-prefix batch decode, bs = 64 requests per GPU (512 over
8 GPUs), GQA 32/8, head_dim 128, page 16.  The shared prefix (8192 tokens) is attended once for the whole
batch by the prefill kernel (all 64 queries of the batch form ONE request of qo_len = 64: the prefix K/V
are read once instead of 64 times), each request's unique suffix (512 tokens) by batch decode, and the two
states are merged (ref: flashinfer/cascade.py:558-795, BatchDecodeWithSharedPrefixPagedKVCacheWrapper).

    python tools/bench_cascade.py                                   # one GPU, no exchange
    torchrun --nproc-per-node N tools/bench_cascade.py              # prefix sequence-sharded over N ranks,
                                                                    # partial states exchanged by all-to-all
"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "flashinfer-ai_amd"))
import torch
import torch.distributed as dist
import flashinfer
from flashinfer import distributed as fdist

world = int(os.environ.get("WORLD_SIZE", "1")); rank = int(os.environ.get("RANK", "0"))
dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0"))); torch.cuda.set_device(dev)
if world > 1:
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("nccl", device_id=dev)

B, HQ, HKV, D, PS = 64, 32, 8, 128, 16
PREFIX, UNIQUE = 8192, 512
g = torch.Generator(device=dev).manual_seed(rank)
# this rank's shard of the shared prefix (sequence-sharded) and its requests' unique suffixes
p_tokens = PREFIX // world
p_pages = p_tokens // PS
u_pages = UNIQUE // PS
cache_p = torch.randn(p_pages, 2, PS, HKV, D, device=dev, dtype=torch.bfloat16, generator=g)
cache_u = torch.randn(B * u_pages, 2, PS, HKV, D, device=dev, dtype=torch.bfloat16, generator=g)
q_local = torch.randn(B, HQ, D, device=dev, dtype=torch.bfloat16, generator=g)
ws1 = torch.zeros(128 << 20, dtype=torch.uint8, device=dev)
ws2 = torch.zeros(128 << 20, dtype=torch.uint8, device=dev)
# prefix: ONE prefill request whose rows are all queries of the whole job (B * world rows)
nq = B * world
pw = flashinfer.BatchPrefillWithPagedKVCacheWrapper(ws1, "NHD")
pw.plan(torch.tensor([0, nq], dtype=torch.int32, device=dev), torch.tensor([0, p_pages], dtype=torch.int32, device=dev),
        torch.arange(p_pages, dtype=torch.int32, device=dev), torch.tensor([PS], dtype=torch.int32, device=dev),
        HQ, HKV, D, PS, causal=False, q_data_type=torch.bfloat16)
dw = flashinfer.BatchDecodeWithPagedKVCacheWrapper(ws2, "NHD")
dw.plan((torch.arange(B + 1, dtype=torch.int32) * u_pages).to(dev),
        torch.randperm(B * u_pages, device=dev, generator=g).to(torch.int32),
        torch.full((B,), PS, dtype=torch.int32, device=dev), HQ, HKV, D, PS, q_data_type=torch.bfloat16)

def step():
    return fdist.sharded_shared_prefix_decode(
        q_local, lambda qa: pw.run(qa, cache_p, return_lse=True), lambda ql: dw.run(ql, cache_u, return_lse=True),
        flashinfer.merge_states, flashinfer.merge_state)

for _ in range(20): step()
torch.cuda.synchronize()
if world > 1: dist.barrier()
t0 = time.perf_counter()
N = 100
for _ in range(N): step()
torch.cuda.synchronize()
if world > 1: dist.barrier()
dt = (time.perf_counter() - t0) / N
# flat (non-cascade) decode would read prefix + suffix for every request
flat_bytes = 2 * B * (PREFIX + UNIQUE) * HKV * D * 2
casc_bytes = 2 * (p_tokens + B * UNIQUE) * HKV * D * 2
if rank == 0:
    print(f"C5 per-GPU share: world={world} bs/GPU={B} prefix={PREFIX} (shard {p_tokens}) unique={UNIQUE}: "
          f"{dt*1e6:.1f} us/step; KV bytes read {casc_bytes/1e6:.1f} MB (flat decode would read {flat_bytes/1e6:.1f} MB "
          f"= {flat_bytes/dt/1e12:.2f} TB/s equivalent); states exchanged per rank: "
          f"{(world-1)*B*HQ*(D*2+4)/1e3 if world>1 else 0:.0f} KB", flush=True)
if world > 1: dist.destroy_process_group()
