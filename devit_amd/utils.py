"""Host-side bookkeeping with the reference's names (utils/dist_utils.py): meters, rank helpers, process-group
init.  Pure host logic; `backend='nccl'` is RCCL on ROCm, `gloo` is used by the CPU tests."""
import datetime
import os
import time
from collections import defaultdict, deque

import torch
import torch.distributed as dist


class SmoothedValue:
    """Windowed / global statistics of a scalar series (utils/dist_utils.py:17-76)."""

    def __init__(self, window_size=20, fmt=None):
        self.fmt = fmt or "{median:.4f} ({global_avg:.4f})"
        self.window = deque(maxlen=window_size)
        self.total, self.count = 0.0, 0

    def update(self, value, n=1):
        self.window.append(value)
        self.count += n
        self.total += value * n

    def synchronize_between_processes(self):
        """Sum count/total over ranks (the window stays local), utils/dist_utils.py:35-46."""
        if not is_dist_avail_and_initialized():
            return
        dev = 'cuda' if dist.get_backend() == 'nccl' else 'cpu'
        t = torch.tensor([self.count, self.total], dtype=torch.float64, device=dev)
        dist.barrier()
        dist.all_reduce(t)
        self.count, self.total = int(t[0].item()), t[1].item()

    median = property(lambda s: torch.tensor(list(s.window)).median().item())
    avg = property(lambda s: torch.tensor(list(s.window), dtype=torch.float32).mean().item())
    global_avg = property(lambda s: s.total / max(s.count, 1))
    max = property(lambda s: max(s.window))
    value = property(lambda s: s.window[-1])

    def __str__(self):
        return self.fmt.format(median=self.median, avg=self.avg, global_avg=self.global_avg, max=self.max,
                               value=self.value)


class MetricLogger:
    def __init__(self, delimiter="\t"):
        self.meters = defaultdict(SmoothedValue)
        self.delimiter = delimiter

    def update(self, **kwargs):
        for k, v in kwargs.items():
            if isinstance(v, torch.Tensor):
                v = v.item()
            self.meters[k].update(v)

    def __getattr__(self, attr):
        if attr in self.__dict__.get("meters", {}):
            return self.meters[attr]
        raise AttributeError(attr)

    def __str__(self):
        return self.delimiter.join(f"{n}: {m}" for n, m in self.meters.items())

    def synchronize_between_processes(self):
        for m in self.meters.values():
            m.synchronize_between_processes()

    def add_meter(self, name, meter):
        self.meters[name] = meter

    def log_every(self, iterable, print_freq, header=''):
        it_time, data_time = SmoothedValue(fmt='{avg:.4f}'), SmoothedValue(fmt='{avg:.4f}')
        start = end = time.time()
        n = len(iterable) if hasattr(iterable, "__len__") else -1
        for i, obj in enumerate(iterable):
            data_time.update(time.time() - end)
            yield obj
            it_time.update(time.time() - end)
            if i % print_freq == 0 or i == n - 1:
                eta = str(datetime.timedelta(seconds=int(it_time.global_avg * (n - i)))) if n > 0 else "?"
                mem = f"  max mem: {torch.cuda.max_memory_allocated() / 2 ** 20:.0f}" if torch.cuda.is_available() else ""
                print(f"{header}  [{i}/{n}]  eta: {eta}  {self}  time: {it_time}  data: {data_time}{mem}")
            end = time.time()
        total = time.time() - start
        print(f"{header} Total time: {datetime.timedelta(seconds=int(total))} ({total / max(n, 1):.4f} s / it)")


def accuracy(output, target, topk=(1,)):
    """timm.utils.accuracy (engine.py:36): top-k hits x 100 / B."""
    maxk = min(max(topk), output.size(1))
    _, pred = output.topk(maxk, 1, True, True)
    hit = pred.t().eq(target.reshape(1, -1).expand_as(pred.t()))
    return [hit[:min(k, maxk)].reshape(-1).float().sum(0) * 100. / target.size(0) for k in topk]


def is_dist_avail_and_initialized():
    return dist.is_available() and dist.is_initialized()


def get_world_size():
    return dist.get_world_size() if is_dist_avail_and_initialized() else 1


def get_rank():
    return dist.get_rank() if is_dist_avail_and_initialized() else 0


def is_main_process():
    return get_rank() == 0


def save_on_master(*args, **kwargs):
    if is_main_process():
        torch.save(*args, **kwargs)


def init_distributed_mode(args):
    """env:// rendezvous from RANK / WORLD_SIZE / LOCAL_RANK (utils/dist_utils.py:215-237)."""
    if 'RANK' in os.environ and 'WORLD_SIZE' in os.environ:
        args.rank, args.world_size = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
        args.gpu = int(os.environ.get('LOCAL_RANK', 0))
    else:
        args.distributed, args.rank, args.world_size, args.gpu = False, 0, 1, 0
        return
    args.distributed = True
    backend = getattr(args, 'dist_backend', None) or ('nccl' if torch.cuda.is_available() else 'gloo')
    if backend == 'nccl':
        torch.cuda.set_device(args.gpu)
    dist.init_process_group(backend=backend, init_method=getattr(args, 'dist_url', 'env://'),
                            world_size=args.world_size, rank=args.rank)
    dist.barrier()
