"""Dev probe: can a worker PROCESS hand CUDA tensors to this one (torch.multiprocessing, hipIpc handles)?"""
import os, sys, time
import torch
import torch.multiprocessing as mp


def worker(q, done):
    torch.cuda.set_device(0)
    for i in range(4):
        t = torch.full((16, 3, 512, 928), float(i), device='cuda')
        torch.cuda.synchronize()
        q.put((i, t))
    done.wait(30)


if __name__ == '__main__':
    print('HSA_ENABLE_IPC_MODE_LEGACY =', os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY'))
    mp.set_start_method('spawn', force=True)
    q, done = mp.Queue(), mp.Event()
    p = mp.Process(target=worker, args=(q, done), daemon=True)
    p.start()
    torch.cuda.set_device(0)
    x = torch.zeros(1, device='cuda')
    t0 = time.time()
    for _ in range(4):
        i, t = q.get(timeout=120)
        t1 = time.time()
        v = float(t.float().mean())
        print(f'batch {i}: mean {v} device {t.device} got after {t1 - t0:.2f} s, read in {(time.time() - t1) * 1e3:.1f} ms', flush=True)
        del t
    done.set()
    p.join(10)
    print('ok')
