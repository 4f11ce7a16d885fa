"""A NumPy stand-in for the slice of the TensorFlow 1.3 Python API that lmb-freiburg/hand3d calls
-- TEST INFRASTRUCTURE ONLY (lives under oracle/, never imported by the product).

Purpose: TensorFlow 1.3 cannot be installed in this environment, but the reference is pure Python on
top of it.  With this package first on sys.path, the reference's OWN modules
(`nets/ColorHandPose3DNetwork.py`, `nets/PosePriorNetwork.py`, `utils/general.py`,
`utils/relative_trafo.py`, `utils/canonical_trafo.py`, `data/BinaryDbReader*.py`) import and run
UNMODIFIED from /root/reference (oracle/refrun.py), evaluated eagerly on NumPy arrays.  That pins
everything the reference itself wrote -- layer order, variable names, concat / flatten order, mask
growth loop, bounding-box and crop-box arithmetic, Rodrigues terms, flip, record layout -- to the
reference's code rather than to a restatement of it.  What stays a restatement is the arithmetic INSIDE
the TensorFlow kernels (conv2d, resize_bilinear, crop_and_resize, dilation2d, softmax ...): those are
the functions of oracle/tf_ops.py and are cross-checked separately
(tests/test_oracle_ops.py).

Execution model: eager by default; since round 3 also a GRAPH facade (bottom of this file): `tf.placeholder` starts a trace,
every op applied to a traced tensor is recorded while it is evaluated on zero-filled stand-in data (which gives the static
shapes the reference's Python asks for), variables without a value yet become variable nodes, and `Session.run(fetches,
feed_dict)` replays the recorded ops on the fed values -- so `scripts/make_tf_fixtures.py`'s real-TensorFlow branch (graph on
placeholders, `net.init(sess)` AFTER the graph is built, `sess.run`) executes here too.

Eager model: a "tensor" is an ndarray subclass that also answers get_shape()/set_shape();
variables come from a process-global store keyed by their full variable-scope name (filled by
tf.contrib.framework.assign_from_values, i.e. by the reference's own `init()`), so `init()` has to be
called BEFORE `inference()` here (TF builds the graph first and assigns afterwards; the order is the
only difference for the caller).  float32 stays float32 throughout (NumPy 2 weak-scalar promotion
matches TF's handling of Python literals).
"""
import contextlib

import numpy as np

from oracle import tf_ops as _T
from oracle import general as _G

float32 = np.float32
float64 = np.float64
int32 = np.int32
int64 = np.int64
uint8 = np.uint8
uint16 = np.uint16
int16 = np.int16
bool = np.bool_          # noqa: A001  (tf.bool)
string = object

__version__ = '1.3.0-numpy-shim'


# ----------------------------------------------------------------------------- tensors
class _Shape(object):
    def __init__(self, dims):
        self._dims = [int(d) for d in dims]

    def as_list(self):
        return list(self._dims)

    def __len__(self):
        return len(self._dims)

    def __getitem__(self, i):
        return self._dims[i]

    def __iter__(self):
        return iter(self._dims)


class Tensor(np.ndarray):
    """ndarray + the two shape methods the reference calls on tf.Tensor (+ the graph facade's node id: None = not traced)."""
    _nid = None

    def __hash__(self):                      # tf.Tensor is hashable by identity (feed_dict keys)
        return id(self)

    def __array_finalize__(self, obj):
        self._nid = None                     # views / results are new tensors: only the recorder tags them

    def __array_ufunc__(self, ufunc, method, *inputs, **kwargs):
        # Python operators on tensors (a * b, -a, a < b ...) arrive here.  Default NumPy behaviour, plus a graph node when traced.
        raw = tuple(i.view(np.ndarray) if isinstance(i, Tensor) else i for i in inputs)
        if 'out' in kwargs:
            kwargs['out'] = tuple(o.view(np.ndarray) if isinstance(o, Tensor) else o for o in kwargs['out'])
        res = getattr(ufunc, method)(*raw, **kwargs)
        if isinstance(res, tuple):
            out = tuple(r.view(Tensor) if isinstance(r, np.ndarray) else r for r in res)
        elif isinstance(res, np.ndarray):
            out = res.view(Tensor)
        elif isinstance(res, np.generic) and method == '__call__':
            out = np.asarray(res).view(Tensor)          # 0-d tensors stay tensors under element-wise ops
        else:
            out = res
        if _graph['on'] and not _graph['busy'] and _any_traced(inputs):
            kw = dict(kwargs)
            _record(lambda *a: getattr(ufunc, method)(*a, **kw), inputs, {}, out)
        return out

    def __getitem__(self, idx):
        out = super(Tensor, self).__getitem__(idx)
        if _graph['on'] and not _graph['busy'] and (self._nid is not None or _any_traced(idx)):
            if not isinstance(out, np.ndarray):
                out = np.asarray(out).view(Tensor)
            _record(lambda a, i: np.ndarray.__getitem__(a.view(np.ndarray), i).view(Tensor) if isinstance(a, np.ndarray) else a[i],
                    (self, idx), {}, out)
        return out

    def get_shape(self):
        return _Shape(self.shape)

    def set_shape(self, shape):
        want = [d for d in shape]
        assert len(want) == self.ndim and all(w is None or int(w) == s for w, s in zip(want, self.shape)), \
            "set_shape(%r) on a tensor of shape %r" % (shape, self.shape)

    @property
    def name(self):
        return 'shim'


def _t(x, dtype=None):
    a = np.asarray(x) if dtype is None else np.asarray(x, dtype=dtype)
    if a.dtype == np.float64 and dtype is None and not isinstance(x, np.ndarray):
        a = a.astype(np.float32)          # Python float literals / lists of them are tf.float32
    if a.dtype == np.int64 and dtype is None and not isinstance(x, (np.ndarray, np.generic)):
        a = a.astype(np.int32)            # Python ints are tf.int32
    return a.view(Tensor)


def _a(x):
    """Input conversion: a NumPy view of a tensor / array / Python literal (TF dtype defaults)."""
    if isinstance(x, np.ndarray):
        return x.view(np.ndarray)
    return _t(x).view(np.ndarray)


def convert_to_tensor(x, dtype=None, name=None):
    return _t(x, dtype)


def constant(value, dtype=None, shape=None, name=None):
    t = _t(value, dtype)
    if shape is not None:
        t = np.broadcast_to(t, shape).copy().view(Tensor)
    return t


def placeholder(dtype, shape=None, name=None):
    """Graph facade: a zero-filled stand-in of the (fully known) shape; switches the recorder on."""
    assert shape is not None and all(d is not None for d in shape), "the graph facade needs fully defined placeholder shapes"
    t = np.zeros([int(d) for d in shape], dtype=dtype).view(Tensor)
    _graph['on'] = True
    t._nid = _new_id()
    _graph['nodes'].append({'kind': 'placeholder', 'nid': t._nid, 'dtype': np.dtype(dtype)})
    return t


# ----------------------------------------------------------------------------- scopes and variables
_scope_stack = []
_variables = {}            # full name (no ':0') -> float32 ndarray


def reset_default_graph():
    del _scope_stack[:]
    _variables.clear()
    _graph.update(on=False, busy=0, next=0)
    del _graph['nodes'][:]


@contextlib.contextmanager
def variable_scope(name, *args, **kwargs):
    _scope_stack.append(name)
    try:
        yield name
    finally:
        _scope_stack.pop()


@contextlib.contextmanager
def name_scope(name, *args, **kwargs):      # does not enter variable names (TF semantics)
    yield name


def constant_initializer(value=0.0, dtype=None):
    return ('constant', value)


def get_variable(name, shape=None, dtype=None, initializer=None, trainable=True, collections=None, **kw):
    full = '/'.join(_scope_stack + [name])
    if full not in _variables and _graph['on'] and shape is not None:
        # graph facade: TF builds the graph first and assigns afterwards (net.init(sess) after inference()): a variable node
        # whose value is read from the store when Session.run replays the graph
        t = np.zeros([int(s) for s in shape], dtype=np.float32).view(Tensor)
        t._nid = _new_id()
        _graph['nodes'].append({'kind': 'variable', 'nid': t._nid, 'name': full, 'shape': [int(s) for s in shape]})
        return t
    if full not in _variables:
        raise KeyError("variable '%s' has no value: call the network's init() with weight files that contain it "
                       "before inference() (eager shim)" % full)
    v = _variables[full]
    if shape is not None:
        assert list(v.shape) == [int(s) for s in shape], \
            "variable %s: stored shape %r, requested %r" % (full, v.shape, list(shape))
    return _t(v)


def global_variables_initializer():
    return None


def check_numerics(tensor, message, name=None):
    a = _a(tensor)
    tracing = _graph['on'] and not _graph['busy'] and _any_traced(tensor)
    if not tracing and not np.all(np.isfinite(a)):
        raise FloatingPointError(message + ' : Tensor had NaN / Inf values')
    return _t(a)


class _Framework(object):
    @staticmethod
    def assign_from_values(var_names_to_values):
        """tf.contrib.framework.assign_from_values: returns (assign_op, feed_dict); Session.run applies it."""
        return ('__assign__', dict(var_names_to_values)), {}


class _Layers(object):
    @staticmethod
    def xavier_initializer_conv2d(*a, **k):
        return ('xavier_conv2d',)

    @staticmethod
    def xavier_initializer(*a, **k):
        return ('xavier',)


class _Contrib(object):
    framework = _Framework()
    layers = _Layers()


contrib = _Contrib()


class ConfigProto(object):
    def __init__(self, *a, **k):
        pass


class GPUOptions(object):
    def __init__(self, *a, **k):
        pass


class Session(object):
    """Eager stand-in: run() applies assign ops and returns already-computed tensors unchanged."""

    def __init__(self, *a, **k):
        pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False

    def close(self):
        pass

    def run(self, fetches, feed_dict=None):
        if _graph['on'] and _any_traced(fetches):
            return _replay(fetches, feed_dict or {})

        def one(f):
            if isinstance(f, tuple) and len(f) == 2 and f[0] == '__assign__':
                for k, v in f[1].items():
                    k = k[:-2] if k.endswith(':0') else k
                    _variables[k] = np.asarray(v, dtype=np.float32)
                return None
            if isinstance(f, dict):
                return {k: one(v) for k, v in f.items()}
            if isinstance(f, (list, tuple)):
                return [one(v) for v in f]
            if f is None:
                return None
            return np.asarray(f).view(np.ndarray)
        return one(fetches)


# ----------------------------------------------------------------------------- element-wise / shape ops
def _un(fn):
    def op(x, name=None):
        return _t(fn(_a(x)))
    return op


sin = _un(np.sin)
cos = _un(np.cos)
sqrt = _un(np.sqrt)
square = _un(np.square)
exp = _un(np.exp)
atan = _un(np.arctan)
floor = _un(np.floor)
is_finite = _un(np.isfinite)
logical_not = _un(np.logical_not)
zeros_like = _un(np.zeros_like)
ones_like = _un(np.ones_like)
identity = _un(lambda a: a.copy())


def round(x, name=None):      # noqa: A001   tf.round = round half to even
    return _t(_T.round_half_even(_a(x)))


def _pair(x, y):
    """TF converts a Python literal / list operand to the dtype of the tensor operand."""
    xa, ya = isinstance(x, np.ndarray), isinstance(y, np.ndarray)
    if xa and not ya:
        return _a(x), np.asarray(y, dtype=x.dtype)
    if ya and not xa:
        return np.asarray(x, dtype=y.dtype), _a(y)
    return _a(x), _a(y)


def _bin(fn):
    def op(x, y, name=None):
        return _t(fn(*_pair(x, y)))
    return op


maximum = _bin(np.maximum)
minimum = _bin(np.minimum)
multiply = _bin(np.multiply)
add = _bin(np.add)
subtract = _bin(np.subtract)
equal = _bin(np.equal)
less = _bin(np.less)
greater = _bin(np.greater)
greater_equal = _bin(np.greater_equal)
logical_and = _bin(np.logical_and)
logical_or = _bin(np.logical_or)


def cast(x, dtype, name=None):
    a = _a(x)
    if np.issubdtype(np.dtype(dtype), np.integer) and np.issubdtype(a.dtype, np.floating):
        with np.errstate(invalid='ignore'):
            return _t(np.trunc(a).astype(dtype))       # float -> int truncates toward zero
    return _t(a.astype(dtype))


def reshape(tensor, shape, name=None):
    return _t(np.reshape(_a(tensor), [int(s) for s in np.asarray(shape).reshape(-1)]))


def expand_dims(x, axis=None, name=None, dim=None):
    return _t(np.expand_dims(_a(x), axis if axis is not None else dim))


def squeeze(x, axis=None, name=None, squeeze_dims=None):
    ax = axis if axis is not None else squeeze_dims
    return _t(np.squeeze(_a(x), axis=None if ax is None else tuple(ax)))


def tile(x, multiples, name=None):
    return _t(np.tile(_a(x), [int(m) for m in multiples]))


def stack(values, axis=0, name=None):
    return _t(np.stack([_a(v) for v in values], axis=axis))


def concat(values, axis, name=None):
    return _t(np.concatenate([_a(v) for v in values], axis=axis))


def transpose(a, perm=None, name=None):
    return _t(np.transpose(_a(a), perm))


def slice(input_, begin, size, name=None):       # noqa: A001
    a = _a(input_)
    idx = tuple(np.s_[int(b):(a.shape[i] if int(s) == -1 else int(b) + int(s))] for i, (b, s) in enumerate(zip(begin, size)))
    return _t(a[idx])


def range(start, limit=None, delta=1, dtype=None, name=None):     # noqa: A001
    if limit is None:
        start, limit = 0, start
    return _t(np.arange(start, limit, delta, dtype=dtype or np.int32))


def ones(shape, dtype=np.float32, name=None):
    return _t(np.ones([int(s) for s in shape], dtype=dtype))


def zeros(shape, dtype=np.float32, name=None):
    return _t(np.zeros([int(s) for s in shape], dtype=dtype))


def where(condition, x=None, y=None, name=None):
    return _t(np.where(_a(condition), *_pair(x, y)))


def one_hot(indices, depth, on_value=1.0, off_value=0.0, axis=-1, dtype=np.float32, name=None):
    idx = _a(indices)
    out = np.full(idx.shape + (int(depth),), off_value, dtype=dtype)
    np.put_along_axis(out, idx[..., None].astype(np.int64), np.asarray(on_value, dtype=dtype), axis=-1)
    return _t(out)


def cond(pred, fn1=None, fn2=None, name=None, true_fn=None, false_fn=None):
    fn1 = fn1 if fn1 is not None else true_fn
    fn2 = fn2 if fn2 is not None else false_fn
    if _graph['on'] and not _graph['busy'] and _any_traced(pred):
        # graph facade: both branches are traced (TF builds both sub-graphs too) and a select node picks at run time; a branch the
        # shim cannot evaluate (training-only ops) becomes an error that fires only if the run selects it
        def branch(fn):
            try:
                return fn()
            except NotImplementedError as e:
                return e
        a, b = branch(fn1), branch(fn2)
        good = a if not isinstance(a, Exception) else b
        out = _t(np.array(_a(good)))

        def select(p, x, y):
            v = x if np.bool_(_a(p)) else y
            if isinstance(v, Exception):
                raise v
            return _t(_a(v))
        _record(select, (pred, a, b), {}, out)
        return out
    return _t(_a(fn1() if np.bool_(_a(pred)) else fn2()))


def boolean_mask(tensor, mask, name=None):
    return _t(_a(tensor)[_a(mask).astype(np.bool_)])


def sparse_to_dense(sparse_indices, output_shape, sparse_values, default_value=0, name=None):
    vals = _a(sparse_values)
    out = np.full([int(s) for s in output_shape], default_value, dtype=vals.dtype)
    ind = _a(sparse_indices).reshape(-1, len(output_shape))
    out[tuple(ind.T)] = vals
    return _t(out)


def dynamic_stitch(indices, data, name=None):
    """merged[indices[m][i, ...]] = data[m][i, ...]"""
    idx = [np.asarray(i).reshape(-1) for i in indices]
    dat = [_a(d) for d in data]
    n = max(int(i.max()) for i in idx) + 1
    rest = dat[0].shape[np.asarray(indices[0]).ndim:]
    out = np.zeros((n,) + tuple(rest), dtype=dat[0].dtype)
    for i, d in zip(idx, dat):
        out[i] = d.reshape((len(i),) + tuple(rest))
    return _t(out)


def matmul(a, b, transpose_a=False, transpose_b=False, name=None):
    a, b = _a(a), _a(b)
    if transpose_a:
        a = np.swapaxes(a, -1, -2)
    if transpose_b:
        b = np.swapaxes(b, -1, -2)
    return _t(np.matmul(a, b))


def matrix_inverse(x, adjoint=False, name=None):
    return _t(np.linalg.inv(_a(x)).astype(_a(x).dtype))


def argmax(x, axis=None, name=None, dimension=None):
    ax = axis if axis is not None else dimension
    return _t(np.argmax(_a(x), axis=ax).astype(np.int64))      # first maximal index


def _empty_identity(kind, dtype):
    if np.issubdtype(dtype, np.floating):
        big = np.inf if _G.EMPTY_REDUCE == 'inf' else np.finfo(dtype).max
    else:
        big = np.iinfo(dtype).max
    return np.asarray(big if kind == 'min' else -big, dtype=dtype)


def _reduce(kind):
    fn = {'min': np.min, 'max': np.max, 'sum': np.sum, 'all': np.all, 'mean': np.mean}[kind]

    def op(x, axis=None, keep_dims=False, name=None, reduction_indices=None):
        a = _a(x)
        ax = axis if axis is not None else reduction_indices
        if kind in ('min', 'max') and a.size == 0:
            # reduce_min / reduce_max of an EMPTY tensor: the reducer's identity (oracle.general.EMPTY_REDUCE)
            return _t(_empty_identity(kind, a.dtype))
        kw = {'dtype': a.dtype} if kind == 'sum' else {}
        return _t(fn(a, axis=None if ax is None else (tuple(ax) if isinstance(ax, (list, tuple)) else int(ax)),
                     keepdims=keep_dims, **kw))
    return op


reduce_min = _reduce('min')
reduce_max = _reduce('max')
reduce_sum = _reduce('sum')
reduce_all = _reduce('all')
reduce_mean = _reduce('mean')


def decode_raw(bytes_, out_type, little_endian=True, name=None):
    return _t(np.frombuffer(bytes_, dtype=np.dtype(out_type).newbyteorder('<' if little_endian else '>')).astype(out_type))


def truncated_normal(*a, **k):
    raise NotImplementedError("training-time augmentation is outside the inference path")


random_uniform = random_crop = truncated_normal


# ----------------------------------------------------------------------------- tf.nn / tf.image
class _NN(object):
    @staticmethod
    def conv2d(input, filter, strides, padding, use_cudnn_on_gpu=None, data_format=None, name=None):   # noqa: A002
        assert padding == 'SAME' and strides[0] == 1 and strides[3] == 1 and strides[1] == strides[2]
        return _t(_T.conv2d_same(_a(input), _a(filter), int(strides[1])))

    @staticmethod
    def bias_add(value, bias, data_format=None, name=None):
        return _t(_T.bias_add(_a(value), _a(bias)))

    @staticmethod
    def max_pool(value, ksize, strides, padding, data_format='NHWC', name=None):
        assert list(ksize) == [1, 2, 2, 1] and list(strides) == [1, 2, 2, 1] and padding == 'VALID'
        return _t(_T.max_pool_2x2(_a(value)))

    @staticmethod
    def avg_pool(value, ksize, strides, padding, data_format='NHWC', name=None):
        assert list(ksize) == [1, 8, 8, 1] and list(strides) == [1, 8, 8, 1] and padding == 'SAME'
        return _t(_T.avg_pool_8x8(_a(value)))

    @staticmethod
    def softmax(logits, dim=-1, name=None):
        assert dim == -1
        return _t(_T.softmax_last(_a(logits)))

    @staticmethod
    def dropout(x, keep_prob, noise_shape=None, seed=None, name=None):
        if float(keep_prob) == 1.0:          # TF 1.3 nn_ops.dropout: "if keep_prob == 1: return x"
            return _t(_a(x))
        raise NotImplementedError("dropout with keep_prob < 1 is training-only (evaluation=False)")

    @staticmethod
    def dilation2d(input, filter, strides, rates, padding, name=None):     # noqa: A002
        x, f = _a(input), _a(filter)
        assert padding == 'SAME' and list(strides) == [1, 1, 1, 1] and list(rates) == [1, 1, 1, 1]
        assert x.ndim == 4 and x.shape[0] == 1 and x.shape[3] == 1 and f.ndim == 3 and f.shape[2] == 1
        if f.shape[0] == f.shape[1] and np.all(f == f.flat[0]):
            out = _T.dilation2d_flat(x[0, :, :, 0], f.shape[0], f.flat[0])     # flat filter: separable, exact
        else:
            out = _T.dilation2d_naive(x[0, :, :, 0], f[:, :, 0])
        return _t(out[None, :, :, None])


nn = _NN()


class _Image(object):
    @staticmethod
    def resize_images(images, size, method=0, align_corners=False):
        assert method == 0 and not align_corners
        a = _a(images)
        h, w = [int(s) for s in np.asarray(size).reshape(-1)]
        if a.ndim == 3:
            return _t(_T.resize_bilinear_legacy(a[None], h, w)[0])
        return _t(_T.resize_bilinear_legacy(a, h, w))

    @staticmethod
    def crop_and_resize(image, boxes, box_ind, crop_size, method='bilinear', extrapolation_value=0, name=None):
        img, bx, bi = _a(image), _a(boxes), _a(box_ind)
        ch, cw = [int(s) for s in _a(crop_size).reshape(-1)]
        return _t(_T.crop_and_resize(img[bi], bx, ch, cw, extrapolation_value))

    @staticmethod
    def random_hue(*a, **k):
        raise NotImplementedError("training-time augmentation is outside the inference path")


image = _Image()


# ----------------------------------------------------------------------------- input pipeline (tf.train, readers)
class _FilenameQueue(object):
    def __init__(self, names):
        self.names = list(names)


class FixedLengthRecordReader(object):
    """Sequential fixed-length records; every read() returns the next record of the file(s), wrapping around
    like a string_input_producer without an epoch limit."""
    _state = {}

    def __init__(self, header_bytes=0, record_bytes=None, footer_bytes=0, name=None):
        self.header, self.rec = int(header_bytes), int(record_bytes)

    def read(self, queue, name=None):
        import os
        fn = queue.names[0]
        key = (fn, self.rec, self.header)
        pos = FixedLengthRecordReader._state.get(key, 0)
        n = (os.path.getsize(fn) - self.header) // self.rec
        with open(fn, 'rb') as f:
            f.seek(self.header + (pos % n) * self.rec)
            value = f.read(self.rec)
        FixedLengthRecordReader._state[key] = pos + 1
        return '%s:%d' % (fn, pos % n), value

    @classmethod
    def rewind(cls):
        cls._state.clear()


class _Train(object):
    @staticmethod
    def string_input_producer(string_tensor, num_epochs=None, shuffle=True, seed=None, capacity=32, name=None):
        return _FilenameQueue(string_tensor)

    @staticmethod
    def batch_join(tensors_list, batch_size, capacity=32, enqueue_many=False, shapes=None, dynamic_pad=False,
                   allow_smaller_final_batch=False, shared_name=None, name=None):
        assert batch_size == 1 and len(tensors_list) == 1, "eager shim: one sample per get()"
        item = tensors_list[0]
        if isinstance(item, dict):
            return {k: _t(_a(v)[None]) for k, v in item.items()}
        return [_t(_a(v)[None]) for v in item]

    @staticmethod
    def shuffle_batch_join(*a, **k):
        raise NotImplementedError("shuffled batching is training-only")

    @staticmethod
    def start_queue_runners(sess=None, coord=None, **k):
        return []


train = _Train()


# ----------------------------------------------------------------------------- graph facade (trace + replay)
# State: `on` after the first tf.placeholder (until reset_default_graph), `busy` > 0 inside a recorded op or a replay (nested
# shim calls are part of the op that is being recorded, not nodes of their own), `nodes` in creation order.
_graph = {'on': False, 'busy': 0, 'next': 0, 'nodes': []}


def _new_id():
    _graph['next'] += 1
    return _graph['next']


def _any_traced(obj):
    if isinstance(obj, Tensor):
        return obj._nid is not None
    if isinstance(obj, (list, tuple)):
        return any(_any_traced(o) for o in obj)
    if isinstance(obj, dict):
        return any(_any_traced(o) for o in obj.values())
    return False


def _tag(out):
    """Node ids for the tensors of an op's result (same nesting); a tensor that already carries one is an input passed through."""
    if isinstance(out, Tensor):
        if out._nid is None:
            out._nid = _new_id()
        return ('t', out._nid)
    if isinstance(out, (list, tuple)):
        return ('l', [_tag(o) for o in out])
    if isinstance(out, dict):
        return ('d', {k: _tag(v) for k, v in out.items()})
    return ('c', None)


def _record(fn, args, kw, out):
    _graph['nodes'].append({'kind': 'op', 'fn': fn, 'args': args, 'kw': kw, 'out': _tag(out)})


def _traced(fn):
    import functools

    @functools.wraps(fn)
    def op(*args, **kw):
        if not _graph['on'] or _graph['busy']:
            return fn(*args, **kw)
        _graph['busy'] += 1
        try:
            out = fn(*args, **kw)
        finally:
            _graph['busy'] -= 1
        if _any_traced(args) or _any_traced(kw):
            _record(fn, args, kw, out)
        return out
    return op


def _replay(fetches, feed):
    env = {}
    for ph, val in feed.items():
        assert isinstance(ph, Tensor) and ph._nid is not None, "feed_dict keys must be placeholders of this graph"
        assert tuple(np.shape(val)) == ph.shape, "fed shape %r, placeholder shape %r" % (np.shape(val), ph.shape)
        env[ph._nid] = np.asarray(val, dtype=ph.dtype).view(Tensor)

    def subst(o):
        if isinstance(o, Tensor) and o._nid is not None:
            if o._nid not in env:
                raise KeyError("graph facade: a placeholder the fetches depend on was not fed")
            return env[o._nid]
        if isinstance(o, list):
            return [subst(v) for v in o]
        if isinstance(o, tuple):
            return tuple(subst(v) for v in o)
        if isinstance(o, dict):
            return {k: subst(v) for k, v in o.items()}
        return o

    def bind(tag, val):
        kind, what = tag
        if kind == 't':
            if what not in env:                       # (an input passed through keeps its value)
                env[what] = val if isinstance(val, Tensor) else np.asarray(val).view(Tensor)
        elif kind == 'l':
            for t, v in zip(what, val):
                bind(t, v)
        elif kind == 'd':
            for k, t in what.items():
                bind(t, val[k])

    _graph['busy'] += 1
    try:
        for n in _graph['nodes']:
            if n['kind'] == 'variable':
                if n['name'] not in _variables:
                    raise KeyError("variable '%s' has no value: net.init(sess, ...) must run before sess.run" % n['name'])
                v = _variables[n['name']]
                assert list(v.shape) == n['shape'], "variable %s: stored shape %r, graph shape %r" % (n['name'], v.shape, n['shape'])
                env[n['nid']] = _t(v)
            elif n['kind'] == 'op':
                try:
                    args, kw = subst(n['args']), subst(n['kw'])
                except KeyError:
                    continue                          # depends on a placeholder that was not fed: only an error if fetched
                bind(n['out'], n['fn'](*args, **kw))

        def fetch(f):
            if isinstance(f, dict):
                return {k: fetch(v) for k, v in f.items()}
            if isinstance(f, (list, tuple)):
                return [fetch(v) for v in f]
            if f is None:
                return None
            return np.asarray(subst(f)).view(np.ndarray)
        return fetch(fetches)
    finally:
        _graph['busy'] -= 1


def _install_tracing():
    import types
    skip = {'placeholder', 'cond', 'get_variable', 'variable_scope', 'name_scope', 'reset_default_graph', 'constant_initializer',
            'global_variables_initializer', 'check_numerics'}
    g = globals()
    for name, f in list(g.items()):
        if isinstance(f, types.FunctionType) and not name.startswith('_') and name not in skip and f.__module__ == __name__:
            g[name] = _traced(f)
    g['check_numerics'] = _traced(g['check_numerics'])
    for cls in (_NN, _Image):
        for name, f in list(vars(cls).items()):
            if isinstance(f, staticmethod):
                setattr(cls, name, staticmethod(_traced(f.__func__)))


_install_tracing()
