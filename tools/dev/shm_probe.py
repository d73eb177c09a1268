"""Dev probe: shared host memory for decoded frames — /dev/shm size, hipHostRegister on a shared mapping, H2D rates."""
import mmap, os, subprocess, tempfile, time
import numpy as np
import torch
print(subprocess.run(['df', '-h', '/dev/shm', '/tmp'], capture_output=True, text=True).stdout)
n = 256 << 20
for where in ('/dev/shm', '/tmp'):
    try:
        fd, path = tempfile.mkstemp(prefix='das_ring_', dir=where)
        os.ftruncate(fd, n)
        mm = mmap.mmap(fd, n)
        a = np.frombuffer(mm, dtype=np.uint8)
        a[:] = 1
        t = torch.from_numpy(a)
        print(where, 'mapped; is_pinned before register:', t.is_pinned())
        rt = torch.cuda.cudart()
        torch.cuda.init()
        err = rt.cudaHostRegister(t.data_ptr(), n, 0)
        print(where, 'cudaHostRegister ->', err, ' is_pinned after:', t.is_pinned())
        d = torch.empty(n, dtype=torch.uint8, device='cuda')
        for nb in (False, True):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(8):
                d[:6 << 20].copy_(t[i * (6 << 20):(i + 1) * (6 << 20)], non_blocking=nb)
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            print(f'  non_blocking={nb}: host {1e3 * (t1 - t0) / 8:.3f} ms per 6 MB copy, total {1e3 * (t2 - t0) / 8:.3f} ms')
        p = torch.empty(6 << 20, dtype=torch.uint8)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(8):
            d[:6 << 20].copy_(p)
        torch.cuda.synchronize()
        print(f'  pageable: {1e3 * (time.perf_counter() - t0) / 8:.3f} ms per 6 MB copy')
        print('  unregister ->', rt.cudaHostUnregister(t.data_ptr()))
        del t, a
        mm.close(); os.close(fd); os.unlink(path)
    except Exception as e:
        print(where, 'FAILED', repr(e))
