import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
print("backend", dist.get_backend())
a = torch.arange(9 * 6 * 8, dtype=torch.float32, device=dev).reshape(9, 6, 8).contiguous()
b = torch.zeros_like(a)
s = torch.cuda.Stream(dev)
ev = torch.cuda.Event(); ev.record(torch.cuda.current_stream(dev)); s.wait_event(ev)
with torch.cuda.stream(s):
    ops = []
    for q in (4, 7, 8):
        ops.append(dist.P2POp(dist.isend, a[q, 1, slice(None)], 0, None, tag=q))
    for q in (2, 5, 6):
        ops.append(dist.P2POp(dist.isend, a[q, 4, slice(1, 8)], 0, None, tag=q))
    for q in (4, 7, 8):
        ops.append(dist.P2POp(dist.irecv, b[q, 5, slice(None)], 0, None, tag=q))
    for q in (2, 5, 6):
        ops.append(dist.P2POp(dist.irecv, b[q, 0, slice(1, 8)], 0, None, tag=q))
    for r in dist.batch_isend_irecv(ops):
        r.wait()
    done = torch.cuda.Event(); done.record(s)
torch.cuda.current_stream(dev).wait_event(done)
torch.cuda.synchronize()
ok = all(torch.equal(b[q, 5], a[q, 1]) for q in (4, 7, 8)) and all(torch.equal(b[q, 0, 1:], a[q, 4, 1:]) for q in (2, 5, 6))
print("self send/recv ok:", ok)
m = torch.tensor([0.25, 0.0], device=dev)
dist.all_reduce(m[:1], op=dist.ReduceOp.MAX)
print("allreduce", m.tolist())
dist.barrier()
dist.destroy_process_group()
