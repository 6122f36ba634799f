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
If a real Ray is present the same classes can be wrapped by `ray.remote` unchanged.
"""
import queue
import threading
from concurrent.futures import Future, wait as _fwait, FIRST_COMPLETED


class _ActorMethod:
    def __init__(self, actor, name):
        self._actor, self._name = actor, name

    def remote(self, *args, **kwargs):
        fut = Future()
        self._actor._mailbox.put((self._name, args, kwargs, fut))
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
        ready = Future()
        self._thread = threading.Thread(target=self._run, args=(cls, args, kwargs, ready), daemon=True)
        self._thread.start()
        ready.result()  # constructor errors surface at .remote(...) creation, like ray.get on first use

    def _run(self, cls, args, kwargs, ready):
        try:
            if self._device is not None:
                import torch
                torch.cuda.set_device(self._device)   # the actor allocates on its creator's GPU, not on cuda:0
            self._obj = cls(*args, **kwargs)
            ready.set_result(True)
        except BaseException as e:  # noqa
            ready.set_exception(e)
            return
        while True:
            item = self._mailbox.get()
            if item is None:
                return
            name, a, k, fut = item
            if not fut.set_running_or_notify_cancel():
                continue
            try:
                fut.set_result(getattr(self._obj, name)(*a, **k))
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
