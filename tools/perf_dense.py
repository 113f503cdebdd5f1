import sys, os, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from oriana_amd._lib import call, ptr, stream_ptr
n, m, K = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
dev = torch.device('cuda')
g = torch.Generator(device=dev).manual_seed(1)
D = torch.rand(n, m, generator=g, device=dev)
X = (torch.rand(n, m, generator=g, device=dev) < 0.1).float()
U = torch.rand(n, K, generator=g, device=dev, dtype=torch.float64)
V = torch.rand(m, K, generator=g, device=dev, dtype=torch.float64)
pi = torch.rand(m, generator=g, device=dev, dtype=torch.float64)
mask = torch.zeros(((n + 31) // 32) * m, dtype=torch.int32, device=dev)
call('oriana_nzmask_f32', ptr(mask), ptr(X), n, m, stream_ptr())
del X
p_d = torch.empty(n, m, dtype=torch.float64, device=dev)
cs = torch.zeros(m, dtype=torch.float64, device=dev)
o1 = torch.zeros(n, K, dtype=torch.float64, device=dev)
o2 = torch.zeros(m, K, dtype=torch.float64, device=dev)
def timeit(f, reps=5):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
which = sys.argv[4] if len(sys.argv) > 4 else 'all'
if which in ('all', 'fused'):
    t = timeit(lambda: call('oriana_dropout_update_fused', ptr(p_d), ptr(D), ptr(U), ptr(V), ptr(pi), ptr(mask), ptr(cs), n, m, K, stream_ptr()))
    print('dropout_fused %.2f ms  (mfma %.1f TF/s, writes %.2f TB/s)' % (t, 2.0 * n * m * K / t / 1e9, 12.0 * n * m / t / 1e9))
if which in ('all', 'dtf'):
    t = timeit(lambda: call('oriana_dense_times_factor', ptr(o1), ptr(D), ptr(V), n, m, K, 0, stream_ptr()))
    print('D V     %.2f ms  (%.1f TF/s useful, D read %.2f TB/s)' % (t, 2.0 * n * m * K / t / 1e9, 4.0 * n * m / t / 1e9))
    t = timeit(lambda: call('oriana_dense_times_factor', ptr(o2), ptr(D), ptr(U), n, m, K, 1, stream_ptr()))
    print('D^T U   %.2f ms  (%.1f TF/s useful, D read %.2f TB/s)' % (t, 2.0 * n * m * K / t / 1e9, 4.0 * n * m / t / 1e9))
