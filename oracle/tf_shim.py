"""ORACLE — TEST INFRASTRUCTURE ONLY (build container only: used by oracle/gen_golden.py).

A small stand-in for the slice of TensorFlow 1.x graph mode that the reference's learner / actor
classes use, evaluated lazily over torch (float64 by default), so that the reference's OWN text —
algos/sac1/{core,actor_learner}.py, example/{core,model}.py, algos/dqn/{core,actor_learner}.py,
algos/sqn/{core,actor_learner}.py — can be *executed* here (TensorFlow is absent from this image)
and its loss / gradient / update values recorded as golden vectors.

What stays "from memory" (the shim IS these definitions; nothing of TensorFlow is available to
check them against) and what therefore the fixtures do NOT pin:
  * tf.layers.dense(x, units, activation): activation(x @ kernel[in, units] + bias[units]);
    variables "<scope>/dense[_k]/kernel", ".../bias" (default-name uniquification per variable
    scope: dense, dense_1, ...; a scope re-entered by name starts its sub-scope counts afresh,
    which is what makes `variable_scope('pi', reuse=True)` find the same variables);
    default initializers glorot-uniform / zeros.
  * tf.train.AdamOptimizer(lr).minimize(loss, var_list): gradients of `loss` with respect to
    `var_list`, then ApplyAdam per variable:  lr_t = lr*sqrt(1-b2^t)/(1-b1^t);
    m += (g-m)*(1-b1); v += (g*g-v)*(1-b2); var -= lr_t*m/(sqrt(v)+eps);  then b1^t *= b1, b2^t *= b2.
    Slots are global variables "<var>/Adam", "<var>/Adam_1" created after every model variable
    (so `zip(get_vars('main'), get_vars('target'))` truncates them away), b^t are "beta1_power".
  * tf.make_template(name, fn, create_scope_now_=True): one captured scope, variables shared by
    every call.
  * ray.experimental.tf_utils.TensorFlowVariables(output, sess): every variable `output` depends
    on (through stop_gradient too — it is a graph edge), in tf.global_variables() order; keys are
    variable names without ":0".
  * Execution order of one sess.run: fetches in list order, each node at most once per run; a node
    created under tf.control_dependencies([...]) runs after those; a variable read after an
    assignment in the same run sees the new value (TF1 reference variables).
Everything else that the fixtures hold — which tensor feeds which loss, the constants, clip,
squash / scale order, which variables each optimizer steps, the polyak pairing — is the
reference's own code running.

tf.random_normal / tf.random.multinomial have no reproducible TF stream: random_normal nodes
take EXPLICIT noise by creation index (Session.noise[i] for the i-th random_normal call of graph
construction); multinomial nodes raise if a fetch ever needs them (no learner fetch does).
"""
import contextlib
import math
import types
from collections import OrderedDict, deque

import numpy as np
import torch

float32 = "float32"
int32 = "int32"
int64 = "int64"

_DTYPE = torch.float64


def set_compute_dtype(dt):
    """torch.float64 (fixtures) or torch.float32 (to show the literal formulation's float32 noise)."""
    global _DTYPE
    _DTYPE = dt


# ------------------------------------------------------------------------------------------------
# graph state
# ------------------------------------------------------------------------------------------------
class _Graph:
    def __init__(self):
        self.variables = []            # creation order == tf.global_variables()
        self.by_name = {}
        self.scope = []                # [(name, reuse)]
        self.scope_count = {}          # full scope name -> times opened (TF's variable_scopes_count)
        self.ctrl = []                 # stack of control-dependency lists
        self.n_random = 0
        self.seed = None
        self.initializer = None        # callable(name, shape) -> ndarray, or None for the default
        self.optimizers = []
        self._default_rs = None

    # -- variable scopes (TF: variable_scope.py _pure_variable_scope / _get_unique_variable_scope)
    def scope_name(self):
        return "/".join(n for n, _ in self.scope)

    def reuse(self):
        return any(r for _, r in self.scope)

    def unique_scope(self, prefix):
        cur = self.scope_name()
        full = cur + "/" + prefix if cur else prefix
        if self.scope_count.get(full, 0) == 0:
            return prefix
        idx = 1
        while self.scope_count.get(full + "_%d" % idx, 0) > 0:
            idx += 1
        return prefix + "_%d" % idx

    @contextlib.contextmanager
    def enter_scope(self, name, reuse=None, restore_counts=False):
        saved = dict(self.scope_count) if restore_counts else None
        self.scope.append((name, bool(reuse)))
        full = self.scope_name()
        self.scope_count[full] = self.scope_count.get(full, 0) + 1
        try:
            yield full
        finally:
            self.scope.pop()
            if restore_counts:
                self.scope_count = saved
            else:
                for k in list(self.scope_count):
                    if k.startswith(full + "/"):
                        self.scope_count[k] = 0

    def default_init(self, name, shape):
        if self._default_rs is None:
            self._default_rs = np.random.RandomState(0 if self.seed is None else self.seed)
        if name.endswith("kernel") and len(shape) == 2:
            lim = math.sqrt(6.0 / (shape[0] + shape[1]))
            return self._default_rs.uniform(-lim, lim, size=shape)
        return np.zeros(shape)

    def get_variable(self, name, shape):
        full = (self.scope_name() + "/" + name) if self.scope else name
        if self.reuse():
            if full not in self.by_name:
                raise ValueError("Variable %s does not exist (reuse=True)" % full)
            v = self.by_name[full]
            assert tuple(v.static_shape) == tuple(shape), (full, v.static_shape, shape)
            return v
        if full in self.by_name:
            raise ValueError("Variable %s already exists (reuse not set)" % full)
        init = (self.initializer or self.default_init)(full, tuple(shape))
        return self.new_variable(full, init)

    def new_variable(self, full, init):
        v = Variable(self, full, torch.as_tensor(np.asarray(init, dtype=np.float64)).to(_DTYPE))
        self.variables.append(v)
        self.by_name[full] = v
        return v


_default_graph = _Graph()
_graph_stack = []


def _g():
    return _graph_stack[-1] if _graph_stack else _default_graph


def reset_default_graph():
    global _default_graph
    _default_graph = _Graph()


class Graph:
    def __init__(self):
        self._g = _Graph()

    @contextlib.contextmanager
    def as_default(self):
        _graph_stack.append(self._g)
        try:
            yield self
        finally:
            _graph_stack.pop()


def get_default_graph():
    return _g()


# ------------------------------------------------------------------------------------------------
# nodes
# ------------------------------------------------------------------------------------------------
class _Shape:
    def __init__(self, dims):
        self.dims = list(dims)

    def as_list(self):
        return list(self.dims)


class Node:
    """A symbolic tensor / op.  fn(*input values) -> torch tensor (or None for a pure op)."""
    __array_ufunc__ = None            # numpy scalars defer to our reflected operators

    def __init__(self, fn, inputs=(), static_shape=None, kind="op", width="inherit"):
        self.graph = _g()
        self.fn, self.inputs, self.kind = fn, list(inputs), kind
        self.ctrl = [c for lst in self.graph.ctrl for c in lst]
        self.static_shape = static_shape
        # last-dimension width, known at graph-construction time (tf.layers.dense sizes its kernel
        # from it): element-wise ops inherit it from whichever operand has one
        if width == "inherit":
            width = static_shape[-1] if static_shape else None
            for i in self.inputs:
                if width is None and isinstance(i, Node):
                    width = i.static_width
        self.static_width = width

    @property
    def shape(self):
        if self.static_shape is None:
            raise NotImplementedError("static shape of a computed tensor is not tracked by the shim")
        return _Shape(self.static_shape)

    @property
    def op(self):
        return self

    def _bin(self, other, f):
        return Node(f, [self, _c(other)])

    def _rbin(self, other, f):
        return Node(f, [_c(other), self])

    __add__ = lambda s, o: s._bin(o, lambda a, b: a + b)
    __radd__ = lambda s, o: s._rbin(o, lambda a, b: a + b)
    __sub__ = lambda s, o: s._bin(o, lambda a, b: a - b)
    __rsub__ = lambda s, o: s._rbin(o, lambda a, b: a - b)
    __mul__ = lambda s, o: s._bin(o, lambda a, b: a * b)
    __rmul__ = lambda s, o: s._rbin(o, lambda a, b: a * b)
    __truediv__ = lambda s, o: s._bin(o, lambda a, b: a / b)
    __rtruediv__ = lambda s, o: s._rbin(o, lambda a, b: a / b)
    __pow__ = lambda s, o: s._bin(o, lambda a, b: a ** b)
    __neg__ = lambda s: Node(lambda a: -a, [s])
    __gt__ = lambda s, o: s._bin(o, lambda a, b: a > b)
    __lt__ = lambda s, o: s._bin(o, lambda a, b: a < b)
    __ge__ = lambda s, o: s._bin(o, lambda a, b: a >= b)
    __le__ = lambda s, o: s._bin(o, lambda a, b: a <= b)
    __hash__ = object.__hash__

    def __bool__(self):
        raise TypeError("a symbolic tensor has no truth value")


def _c(x):
    """Python / NumPy numbers become constants of the compute dtype (TF: of the tensor's dtype)."""
    if isinstance(x, Node):
        return x
    val = torch.as_tensor(np.asarray(x, dtype=np.float64)).to(_DTYPE)
    return Node(lambda: val, [], kind="const")


class Variable(Node):
    def __init__(self, graph, full_name, value):
        Node.__init__(self, None, [], static_shape=tuple(value.shape), kind="variable")
        self.graph = graph
        self.ctrl = []
        self.name = full_name + ":0"
        self.op_name = full_name
        self.value = value

    def assign_value(self, new):
        self.value = torch.as_tensor(np.asarray(new, dtype=np.float64)).to(_DTYPE).reshape(self.value.shape).clone()


class _Run:
    def __init__(self, feed, noise):
        self.memo, self.feed, self.noise = {}, feed, noise
        self.noise_used = []

    def eval(self, node):
        if not isinstance(node, Node):
            return node
        if node in self.memo:
            return self.memo[node]
        for c in node.ctrl:
            self.eval(c)
        if node.kind == "variable":
            val = node.value.detach().clone().requires_grad_(True)   # this run's leaf for autograd
        elif node.kind == "placeholder":
            if node not in self.feed:
                raise KeyError("placeholder not fed")
            val = torch.as_tensor(np.asarray(self.feed[node], dtype=np.float64)).to(_DTYPE)
            want = node.static_shape
            assert val.dim() == len(want) and all(w is None or w == s for w, s in zip(want, val.shape)), \
                (tuple(val.shape), want)
        elif node.kind == "run":                                   # needs the run itself (optimizers, noise)
            val = node.fn(self)
        else:
            val = node.fn(*[self.eval(i) for i in node.inputs])
        self.memo[node] = val
        return val

    def written(self, var):
        self.memo.pop(var, None)


def placeholder(dtype=None, shape=None):
    return Node(None, [], static_shape=tuple(shape), kind="placeholder")


# ------------------------------------------------------------------------------------------------
# ops used by the reference
# ------------------------------------------------------------------------------------------------
def _un(f):
    return lambda x, name=None: Node(f, [_c(x)])


exp = _un(torch.exp)
log = _un(torch.log)
tanh = _un(torch.tanh)
identity = _un(lambda a: a)
stop_gradient = _un(lambda a: a.detach())


def minimum(a, b):
    return Node(torch.minimum, [_c(a), _c(b)])


def reduce_sum(x, axis=None):
    return Node((lambda a: a.sum()) if axis is None else (lambda a: a.sum(dim=axis)), [x], width=None)


def reduce_mean(x, axis=None):
    return Node((lambda a: a.mean()) if axis is None else (lambda a: a.mean(dim=axis)), [x], width=None)


def concat(values, axis=-1):
    ws = [v.static_width for v in values]
    return Node(lambda *v: torch.cat(v, dim=axis), list(values),
                width=sum(ws) if axis in (-1, 1) and all(w is not None for w in ws) else None)


def squeeze(x, axis=None):
    return Node(lambda a: a.squeeze(axis), [x], width=None)


def cast(x, dtype):
    if dtype == float32:
        return Node(lambda a: a.to(_DTYPE), [x])
    if dtype in (int32, int64):
        return Node(lambda a: a.to(torch.int64), [x])       # float -> int truncates toward zero, as tf.cast
    raise NotImplementedError(dtype)


def shape(x):
    return Node(lambda a: tuple(a.shape), [x])


def one_hot(indices, depth):
    return Node(lambda i: torch.nn.functional.one_hot(i.to(torch.int64), depth).to(_DTYPE), [indices])


def argmax(x, axis=None):
    return Node(lambda a: torch.argmax(a, dim=axis), [x], width=None)    # first maximal index, as tf.argmax


def random_normal(shp, *a, **k):
    g = _g()
    idx = g.n_random
    g.n_random += 1

    def fn(run):
        want = run.eval(shp) if isinstance(shp, Node) else tuple(shp)
        if run.noise is None or idx >= len(run.noise) or run.noise[idx] is None:
            raise RuntimeError("tf.random_normal call #%d has no explicit noise" % idx)
        val = torch.as_tensor(np.asarray(run.noise[idx], dtype=np.float64)).to(_DTYPE)
        assert tuple(val.shape) == tuple(want), (idx, tuple(val.shape), want)
        run.noise_used.append(idx)
        return val
    return Node(fn, [shp] if isinstance(shp, Node) else [], kind="run")


def _multinomial(logits, num_samples, **k):
    def fn(run):
        raise RuntimeError("tf.random.multinomial evaluated: no reproducible stream in the shim")
    return Node(fn, [logits], kind="run")


def group(*inputs, **k):
    flat = []
    for i in inputs:
        flat += list(i) if isinstance(i, (list, tuple)) else [i]
    return Node(lambda *v: None, flat)


def assign(ref, value):
    def fn(run):
        new = run.eval(_c(value))
        ref.value = new.detach().clone()
        run.written(ref)
        return ref.value
    return Node(fn, [_c(value), ref], kind="run")


@contextlib.contextmanager
def control_dependencies(ops):
    g = _g()
    g.ctrl.append(list(ops))
    try:
        yield
    finally:
        g.ctrl.pop()


@contextlib.contextmanager
def variable_scope(name_or_scope, default_name=None, reuse=None):
    g = _g()
    name = name_or_scope if name_or_scope is not None else g.unique_scope(default_name)
    with g.enter_scope(name, reuse) as full:
        yield full


def make_template(name, func, create_scope_now_=False, **kw):
    g = _g()
    assert create_scope_now_, "only create_scope_now_=True is used by the reference"
    uniq = g.unique_scope(name)
    with g.enter_scope(uniq):
        prefix = list(g.scope)
    state = {"first": True}

    def call(*a, **k):
        saved = g.scope
        g.scope = list(prefix[:-1])
        try:
            with g.enter_scope(uniq, reuse=not state["first"], restore_counts=True):
                out = func(*a, **k)
        finally:
            g.scope = saved
        state["first"] = False
        return out
    return call


def get_variable(name, dtype=None, initializer=None, shape=None):
    raise NotImplementedError("tf.get_variable (alpha='auto') is outside the hot path")


def global_variables():
    return list(_g().variables)


def global_variables_initializer():
    return Node(lambda: None, [])


def set_random_seed(seed):
    _g().seed = seed


def _dense(x, units, activation=None, **k):
    g = _g()
    if x.static_width is None:
        raise NotImplementedError("dense() on an input of unknown static width")
    with variable_scope(None, default_name="dense"):
        kernel = g.get_variable("kernel", (x.static_width, units))
        bias = g.get_variable("bias", (units,))
    out = Node(lambda a, w, b: a @ w + b, [x, kernel, bias], width=units)
    return out if activation is None else activation(out)


def _relu(x, name=None):
    return Node(torch.relu, [x])


def _log_softmax(x, axis=-1):
    return Node(lambda a: torch.log_softmax(a, dim=axis), [x])


# ------------------------------------------------------------------------------------------------
# optimizer
# ------------------------------------------------------------------------------------------------
class AdamOptimizer:
    def __init__(self, learning_rate=0.001, beta1=0.9, beta2=0.999, epsilon=1e-8, name="Adam"):
        self.lr, self.beta1, self.beta2, self.epsilon, self.name = learning_rate, beta1, beta2, epsilon, name
        self.last_grads = None
        self.var_list = None
        _g().optimizers.append(self)

    def minimize(self, loss, var_list=None):
        g = _g()
        assert var_list, "the reference always passes var_list"
        self.var_list = list(var_list)

        def uniq(base):
            if base not in g.by_name:
                return base
            i = 1
            while "%s_%d" % (base, i) in g.by_name:
                i += 1
            return "%s_%d" % (base, i)
        self.b1p = g.new_variable(uniq("beta1_power"), self.beta1)
        self.b2p = g.new_variable(uniq("beta2_power"), self.beta2)
        self.m, self.v = {}, {}
        for var in self.var_list:
            self.m[var] = g.new_variable(uniq(var.op_name + "/" + self.name), np.zeros(var.static_shape))
            self.v[var] = g.new_variable(uniq(var.op_name + "/" + self.name), np.zeros(var.static_shape))

        def fn(run):
            loss_v = run.eval(loss)
            leaves = [run.eval(v) for v in self.var_list]
            grads = torch.autograd.grad(loss_v, leaves, retain_graph=True, allow_unused=True)
            self.last_grads = OrderedDict()
            one = torch.tensor(1.0, dtype=_DTYPE)
            lr, b1, b2, eps = (torch.tensor(x, dtype=_DTYPE) for x in (self.lr, self.beta1, self.beta2, self.epsilon))
            lr_t = lr * torch.sqrt(one - self.b2p.value) / (one - self.b1p.value)
            for var, gr in zip(self.var_list, grads):
                if gr is None:
                    continue
                gr = gr.detach()
                self.last_grads[var.op_name] = gr.clone()
                m, v = self.m[var], self.v[var]
                m.value = m.value + (gr - m.value) * (one - b1)
                v.value = v.value + (gr * gr - v.value) * (one - b2)
                var.value = var.value - (m.value * lr_t) / (torch.sqrt(v.value) + eps)
                run.written(var)
            self.b1p.value = self.b1p.value * b1
            self.b2p.value = self.b2p.value * b2
            return None
        return Node(fn, [loss] + self.var_list, kind="run")


# ------------------------------------------------------------------------------------------------
# session
# ------------------------------------------------------------------------------------------------
class _Proto:
    def __init__(self, **k):
        self.__dict__.update(k)

    def __getattr__(self, name):
        sub = _Proto()
        self.__dict__[name] = sub
        return sub


def ConfigProto(**k):
    return _Proto(**k)


class Session:
    def __init__(self, config=None, graph=None):
        self.graph = _g()
        self.noise = None             # explicit normals by random_normal creation index
        self.last_outputs = None
        self.last_noise_used = None

    def run(self, fetches, feed_dict=None):
        run = _Run(dict(feed_dict or {}), self.noise)
        single = not isinstance(fetches, (list, tuple))
        outs = []
        for f in ([fetches] if single else fetches):
            v = run.eval(f)
            if torch.is_tensor(v):
                v = v.detach().numpy().copy()
            outs.append(v)
        self.last_outputs = outs
        self.last_noise_used = sorted(run.noise_used)
        return outs[0] if single else outs


# ------------------------------------------------------------------------------------------------
# ray.experimental.tf_utils.TensorFlowVariables
# ------------------------------------------------------------------------------------------------
class TensorFlowVariables:
    def __init__(self, output, sess=None, input_variables=None):
        self.sess = sess
        outputs = list(output) if isinstance(output, (list, tuple)) else [output]
        queue, seen, names = deque(outputs), set(outputs), set()
        while queue:
            n = queue.popleft()
            if not isinstance(n, Node):
                continue
            for i in list(n.inputs) + list(n.ctrl):
                if isinstance(i, Node) and i not in seen:
                    seen.add(i)
                    queue.append(i)
            if n.kind == "variable":
                names.add(n.op_name)
        g = outputs[0].graph
        lst = [v for v in g.variables if v.op_name in names]
        if input_variables is not None:
            lst += list(input_variables)
        self.variables = OrderedDict((v.op_name, v) for v in lst)

    def get_weights(self):
        return {k: v.value.detach().numpy().copy() for k, v in self.variables.items()}

    def set_weights(self, new_weights):
        for k, val in new_weights.items():
            if k in self.variables:
                self.variables[k].assign_value(val)


# ------------------------------------------------------------------------------------------------
# module objects to put into sys.modules
# ------------------------------------------------------------------------------------------------
def _inert(*a, **k):
    return None


class _FileWriter:
    def __init__(self, *a, **k):
        pass

    def add_summary(self, *a, **k):
        pass

    def flush(self):
        pass


def _tf_variable(initial_value, *a, **k):
    g = _g()
    name = g.unique_scope("Variable")
    with g.enter_scope(name):
        full = g.scope_name()
    return g.new_variable(full, initial_value)


def as_modules():
    """-> (tensorflow stand-in, ray.experimental.tf_utils stand-in)."""
    tf = types.ModuleType("tensorflow")
    for k in ("float32", "int32", "int64", "placeholder", "exp", "log", "tanh", "identity", "stop_gradient",
              "minimum", "reduce_sum", "reduce_mean", "concat", "squeeze", "cast", "shape", "one_hot", "argmax",
              "random_normal", "group", "assign", "control_dependencies", "variable_scope", "make_template",
              "get_variable", "global_variables", "global_variables_initializer", "set_random_seed", "Graph",
              "Session", "ConfigProto", "reset_default_graph", "get_default_graph"):
        setattr(tf, k, globals()[k])
    tf.Variable = _tf_variable
    tf.layers = types.SimpleNamespace(dense=_dense)
    tf.nn = types.SimpleNamespace(relu=_relu, log_softmax=_log_softmax, tanh=tanh)
    tf.train = types.SimpleNamespace(AdamOptimizer=AdamOptimizer)
    tf.random = types.SimpleNamespace(multinomial=_multinomial, normal=random_normal)
    tf.summary = types.SimpleNamespace(scalar=_inert, merge=_inert, FileWriter=_FileWriter)
    tf.contrib = types.SimpleNamespace(framework=types.SimpleNamespace(get_variables_to_restore=global_variables))
    tf.app = types.SimpleNamespace(flags=types.SimpleNamespace(
        FLAGS=types.SimpleNamespace(), DEFINE_string=_inert, DEFINE_integer=_inert, DEFINE_float=_inert,
        DEFINE_boolean=_inert))
    tfu = types.ModuleType("ray.experimental.tf_utils")
    tfu.TensorFlowVariables = TensorFlowVariables
    return tf, tfu
