"""A minimal stand-in for the slice of Ray's API the reference's drivers use
(example/dsac.py:14,51,76,133,153,218-238):

    @remote / @remote(num_gpus=1, max_calls=1)      on classes (actors) and functions (tasks)
    Handle = Class.remote(*ctor_args);  fut = handle.method.remote(*args)
    fut = task.remote(*args);  get(fut) / get([futs]);  wait([futs]);  init()

Ray is not installed in this image (and the hot path no longer crosses process boundaries:
replay, learner and environments live in the same device's HBM).  Semantics kept:
  * an actor executes its methods serially in arrival order (one mailbox thread per actor —
    algos/sac1/sac_ray.py:316-317 relies on this);
  * `.remote()` returns immediately with a future; `store.remote(...)` may be fire-and-forget;
  * tasks run concurrently (one thread each); exceptions surface at `get`.
  * a Ray actor is a process of its own: its device work never queues behind its callers'.  Here every actor thread issues its
    device work on a HIP stream of its own (non-blocking), so a replay actor's `sample_batch` (gather + copy down) runs beside
    the learner's update instead of behind it on the shared default stream — what lets the reference's `Cache` prefetch
    (algos/sac1/sac1.py:103-130) hide the sample behind the train on the host-buffer surface.  Arguments and results cross as
    in Ray, by value: a CUDA tensor handed to an actor is waited for (event) before the method runs, and an actor drains its
    stream before a result that holds CUDA tensors is published.
If a real Ray is present the same classes can be wrapped by `ray.remote` unchanged.
"""
import queue
import threading
from concurrent.futures import Future, wait as _fwait, FIRST_COMPLETED


def _has_cuda(x, depth=0):
    """Does a (shallowly nested) argument / result hold a CUDA tensor?"""
    if hasattr(x, "is_cuda"):
        return bool(x.is_cuda)
    if depth < 2:
        if isinstance(x, (list, tuple)):
            return any(_has_cuda(v, depth + 1) for v in x)
        if isinstance(x, dict):
            return any(_has_cuda(v, depth + 1) for v in x.values())
    return False


def _record_default(x, depth=0):
    """Device tensors an actor hands out were allocated on ITS stream; the caller most likely reads them on the default stream:
    tell the caching allocator, so that the block is not given out again before that stream has passed the reads."""
    import torch
    if hasattr(x, "is_cuda"):
        if x.is_cuda:
            x.record_stream(torch.cuda.default_stream(x.device))
    elif depth < 2:
        for v in (x.values() if isinstance(x, dict) else x if isinstance(x, (list, tuple)) else ()):
            _record_default(v, depth + 1)


class _ActorMethod:
    def __init__(self, actor, name):
        self._actor, self._name = actor, name

    def remote(self, *args, **kwargs):
        fut = Future()
        ev = None
        if self._actor._stream is not None and (_has_cuda(args) or _has_cuda(kwargs)):
            import torch
            ev = torch.cuda.Event()
            ev.record()                        # the caller's stream, as far as it has been issued: the actor's stream waits for it
        self._actor._mailbox.put((self._name, args, kwargs, fut, ev))
        return fut


def _current_device():
    try:
        import torch
        return torch.cuda.current_device() if torch.cuda.is_available() else None
    except Exception:  # noqa
        return None


class ActorHandle:
    def __init__(self, cls, args, kwargs):
        self._mailbox = queue.Queue()
        self._obj = None
        self._device = _current_device()   # torch's current device is thread-local and starts at 0 in a new thread
        self._stream = None
        ready = Future()
        self._thread = threading.Thread(target=self._run, args=(cls, args, kwargs, ready), daemon=True)
        self._thread.start()
        ready.result()  # constructor errors surface at .remote(...) creation, like ray.get on first use

    def _run(self, cls, args, kwargs, ready):
        try:
            if self._device is not None:
                import torch
                torch.cuda.set_device(self._device)   # the actor allocates on its creator's GPU, not on cuda:0
                self._stream = torch.cuda.Stream()    # (non-blocking: no implicit ordering against the default stream)
                torch.cuda.set_stream(self._stream)
            self._obj = cls(*args, **kwargs)
            if self._stream is not None:
                self._stream.synchronize()            # what the constructor queued (ring fill, parameter upload) has landed
            ready.set_result(True)
        except BaseException as e:  # noqa
            ready.set_exception(e)
            return
        while True:
            item = self._mailbox.get()
            if item is None:
                return
            name, a, k, fut, ev = item
            if not fut.set_running_or_notify_cancel():
                continue
            try:
                if ev is not None:
                    self._stream.wait_event(ev)
                res = getattr(self._obj, name)(*a, **k)
                if self._stream is not None and _has_cuda(res):
                    self._stream.synchronize()
                    _record_default(res)
                fut.set_result(res)
            except BaseException as e:  # noqa
                fut.set_exception(e)

    def __getattr__(self, name):
        if name.startswith("_"):
            raise AttributeError(name)
        return _ActorMethod(self, name)

    def _stop(self):
        self._mailbox.put(None)


class _RemoteClass:
    def __init__(self, cls):
        self._cls = cls

    def remote(self, *args, **kwargs):
        return ActorHandle(self._cls, args, kwargs)

    def __call__(self, *args, **kwargs):  # plain construction still works
        return self._cls(*args, **kwargs)


class _RemoteFunction:
    def __init__(self, fn):
        self._fn = fn

    def remote(self, *args, **kwargs):
        fut = Future()
        device = _current_device()

        def run():
            if not fut.set_running_or_notify_cancel():
                return
            try:
                if device is not None:
                    import torch
                    torch.cuda.set_device(device)
                fut.set_result(self._fn(*args, **kwargs))
            except BaseException as e:  # noqa
                fut.set_exception(e)
        threading.Thread(target=run, daemon=True).start()
        return fut

    def __call__(self, *args, **kwargs):
        return self._fn(*args, **kwargs)


def remote(*args, **kwargs):
    """`@remote` and `@remote(num_gpus=1, max_calls=1)` (resource hints are accepted and ignored:
    placement is one process per GPU, decided by the launcher)."""
    def wrap(obj):
        return _RemoteClass(obj) if isinstance(obj, type) else _RemoteFunction(obj)
    if len(args) == 1 and not kwargs and (callable(args[0]) or isinstance(args[0], type)):
        return wrap(args[0])
    return wrap


def get(x, timeout=None):
    if isinstance(x, (list, tuple)):
        return [get(f, timeout) for f in x]
    return x.result(timeout) if isinstance(x, Future) else x


def wait(futures, num_returns=1, timeout=None):
    futures = list(futures)
    done = set()
    while len(done) < num_returns:
        d, _ = _fwait([f for f in futures if f not in done], timeout=timeout, return_when=FIRST_COMPLETED)
        if not d:
            break
        done |= d
    ready = [f for f in futures if f in done]
    return ready, [f for f in futures if f not in done]


def init(*args, **kwargs):
    return None
