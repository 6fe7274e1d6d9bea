"""Does this RCCL build accept two ranks on ONE GPU?  (NCCL refuses: "Duplicate GPU detected".)  If it does, the 2-rank RCCL tests
can run on the 1-GPU boxes.  RESULT (round 3, one gpurun call): both ranks exit 1 at the first collective -- torch's librccl.so carries
NCCL's "Duplicate GPU detected" check and no multi-rank-per-GPU switch, so RCCL with more than one rank stays unexercised until a
box has two GPUs (tests/test_joint_forward_gpu.py::test_two_rank_rccl_on_two_gpus then runs instead of skipping).  Run under: python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 ... this file"""
import os, sys, torch, torch.distributed as dist
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda:0"))
x = torch.full((1024,), float(rank + 1), device="cuda")
dist.all_reduce(x)
torch.cuda.synchronize()
print(f"rank {rank}: all_reduce -> {x[0].item()} (expect {world * (world + 1) / 2})", flush=True)
# uneven all-to-all (the head exchange)
send = torch.arange(5, device="cuda", dtype=torch.float32) + 10 * rank
out = torch.empty(3 if rank == 0 else 7, device="cuda") if world == 2 else None
if world == 2:
    in_split = [2, 3]
    out_split = [2, 2] if rank == 0 else [3, 3]
    out = torch.empty(sum(out_split), device="cuda")
    dist.all_to_all_single(out, send, out_split, in_split)
    torch.cuda.synchronize()
    print(f"rank {rank}: all_to_all_single -> {out.tolist()}", flush=True)
dist.destroy_process_group()
