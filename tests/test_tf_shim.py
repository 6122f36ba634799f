"""oracle/tf_shim.py on its own (no reference needed): the TF1 behaviours the golden math fixtures lean on — variable naming and
reuse, the collection order, Adam slot variables and the zip truncation they cause, sess.run ordering with control dependencies,
ApplyAdam on a scalar, explicit noise by creation index, TensorFlowVariables' dependency walk."""
import math

import numpy as np
import pytest

from oracle import tf_shim


@pytest.fixture()
def tf():
    tf_shim.reset_default_graph()
    tf_shim.set_compute_dtype(tf_shim.torch.float64)
    mod, _ = tf_shim.as_modules()
    return mod


def names(tf):
    return [v.name for v in tf.global_variables()]


def test_variable_names_reuse_and_template(tf):
    x = tf.placeholder(tf.float32, shape=(None, 3))
    with tf.variable_scope("main"):
        with tf.variable_scope("pi"):
            h = tf.layers.dense(x, units=5, activation=tf.nn.relu)
            a = tf.layers.dense(h, units=2)
            b = tf.layers.dense(h, units=2, activation=tf.tanh)
        with tf.variable_scope("pi", reuse=True):                      # the second policy() call of core.py:99-101
            h2 = tf.layers.dense(x, units=5, activation=tf.nn.relu)
            tf.layers.dense(h2, units=2)
        q_tp = tf.make_template("q1", lambda z: tf.layers.dense(tf.layers.dense(z, units=4, activation=tf.nn.relu), units=1), create_scope_now_=True)
        q_tp(x)
        q_tp(x)                                                         # same variables again (algos/dqn/core.py:44-48)
    with tf.variable_scope("target"):
        with tf.variable_scope("pi"):
            tf.layers.dense(x, units=5)
    assert names(tf) == ["main/pi/dense/kernel:0", "main/pi/dense/bias:0", "main/pi/dense_1/kernel:0", "main/pi/dense_1/bias:0",
                         "main/pi/dense_2/kernel:0", "main/pi/dense_2/bias:0", "main/q1/dense/kernel:0", "main/q1/dense/bias:0",
                         "main/q1/dense_1/kernel:0", "main/q1/dense_1/bias:0", "target/pi/dense/kernel:0", "target/pi/dense/bias:0"]
    with pytest.raises(ValueError):
        with tf.variable_scope("main"):
            with tf.variable_scope("pi"):
                tf.layers.dense(x, units=5)                             # exists, reuse not set
    sess = tf.Session()
    out = sess.run([a, b], {x: np.ones((4, 3))})
    assert out[0].shape == (4, 2) and np.abs(out[1]).max() <= 1.0


def test_adam_slots_are_global_variables_created_after_the_model(tf):
    """Why `zip(get_vars('main'), get_vars('target'))` pairs model variables only (actor_learner.py:85-87)."""
    x = tf.placeholder(tf.float32, shape=(None, 2))
    with tf.variable_scope("main"):
        y = tf.layers.dense(x, units=1)
    with tf.variable_scope("target"):
        tf.layers.dense(x, units=1)
    loss = tf.reduce_mean(y ** 2)
    main_vars = [v for v in tf.global_variables() if "main" in v.name]
    tf.train.AdamOptimizer(learning_rate=0.1).minimize(loss, var_list=main_vars)
    after = [v.name for v in tf.global_variables() if "main" in v.name]
    assert after == ["main/dense/kernel:0", "main/dense/bias:0", "main/dense/kernel/Adam:0", "main/dense/kernel/Adam_1:0",
                     "main/dense/bias/Adam:0", "main/dense/bias/Adam_1:0"]
    targ = [v.name for v in tf.global_variables() if "target" in v.name]
    assert [m for m, _ in zip(after, targ)] == ["main/dense/kernel:0", "main/dense/bias:0"]
    assert "beta1_power:0" in names(tf) and "beta2_power:0" in names(tf)


def test_run_order_control_dependencies_and_apply_adam(tf):
    """One variable w, loss = mean((w x)^2): forward value from the pre-update variable, then Adam, then a polyak-style assign that
    sees the POST-update value because it is built under control_dependencies — three runs against the closed form."""
    x = tf.placeholder(tf.float32, shape=(None, 1))
    tf_shim.get_default_graph().initializer = lambda name, shape: np.full(shape, 0.5 if "main" in name else 2.0)
    with tf.variable_scope("main"):
        y = tf.layers.dense(x, units=1)
    with tf.variable_scope("target"):
        tf.layers.dense(x, units=1)
    loss = tf.reduce_mean(y ** 2)
    mv = [v for v in tf.global_variables() if "main" in v.name]
    tv = [v for v in tf.global_variables() if "target" in v.name]
    train = tf.train.AdamOptimizer(learning_rate=0.1).minimize(loss, var_list=mv)
    with tf.control_dependencies([train]):
        upd = tf.group([tf.assign(t, 0.9 * t + (1 - 0.9) * m) for m, t in zip(mv, tv)])
    sess = tf.Session()
    w = b = 0.5
    tw = 2.0
    m1 = v1 = mb = vb = 0.0
    for t in range(1, 4):
        xs = np.array([[1.0], [2.0], [-1.0]]) * t
        got = sess.run([loss, train, upd], {x: xs})
        pre = w * xs[:, 0] + b
        assert abs(got[0] - np.mean(pre ** 2)) < 1e-12                       # the fetch saw the pre-update variable
        gw, gb = np.mean(2 * pre * xs[:, 0]), np.mean(2 * pre)
        lr_t = 0.1 * math.sqrt(1 - 0.999 ** t) / (1 - 0.9 ** t)
        m1 = 0.9 * m1 + 0.1 * gw; v1 = 0.999 * v1 + 0.001 * gw * gw
        mb = 0.9 * mb + 0.1 * gb; vb = 0.999 * vb + 0.001 * gb * gb
        w -= lr_t * m1 / (math.sqrt(v1) + 1e-8); b -= lr_t * mb / (math.sqrt(vb) + 1e-8)
        tw = 0.9 * tw + 0.1 * w                                             # polyak with the post-update main
        assert abs(float(mv[0].value) - w) < 1e-12 and abs(float(tv[0].value) - tw) < 1e-12


def test_explicit_noise_by_creation_index_and_stop_gradient(tf):
    x = tf.placeholder(tf.float32, shape=(None, 2))
    n0 = tf.random_normal(tf.shape(x))
    n1 = tf.random_normal(tf.shape(x))
    out = tf.stop_gradient(x * 2) + n1            # fetching this evaluates draw #1 only
    sess = tf.Session()
    sess.noise = [None, np.full((3, 2), 7.0)]
    got = sess.run(out, {x: np.ones((3, 2))})
    assert (got == 9.0).all() and sess.last_noise_used == [1]
    with pytest.raises(RuntimeError):
        sess.run(n0, {x: np.ones((3, 2))})


def test_tensorflow_variables_walks_through_stop_gradient(tf):
    _, tfu = tf_shim.as_modules()
    x = tf.placeholder(tf.float32, shape=(None, 2))
    with tf.variable_scope("main"):
        with tf.variable_scope("q"):
            q = tf.layers.dense(x, units=1)
        with tf.variable_scope("pi"):
            p = tf.layers.dense(x, units=1)
    with tf.variable_scope("target"):
        t = tf.layers.dense(x, units=1)
    loss = tf.reduce_mean((tf.stop_gradient(t + p) - q) ** 2)
    v = tfu.TensorFlowVariables(loss, tf.Session())
    assert list(v.variables) == ["main/q/dense/kernel", "main/q/dense/bias", "main/pi/dense/kernel", "main/pi/dense/bias",
                                 "target/dense/kernel", "target/dense/bias"]
    v.set_weights({"main/q/dense/bias": np.array([3.0]), "not/a/variable": np.zeros(1)})
    assert v.get_weights()["main/q/dense/bias"][0] == 3.0
