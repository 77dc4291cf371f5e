"""Tree-walking evaluator for the Rust subset (see __init__.py).

Arithmetic model
  f32   numpy.float32 scalars (or 1-D float32 arrays: a batch of independent lanes run in lockstep); every operation is
        one IEEE-754 binary32 operation, i.e. exactly one rounding -- Rust never contracts a*b+c.
  f64   Python floats (binary64).  cos/sin/tan/exp/pow/... go to the C library through Python's math module (f64) and
        ctypes (f32: sinf, cosf, powf, ...): the same glibc libm a Rust binary on this machine calls.
  ints  arbitrary-precision values tagged with their Rust type; results wrap to the type's width like a release
        build (every implicit wrap is counted in Interp.overflows so a fixture run can assert there were none).
  Literals without a suffix stay untyped until an operation, a declared type (let / parameter / field / return) or the
  value they overwrite gives them one -- a dynamic stand-in for Rust's inference; left alone they default to i32 / f64.

Value model
  Aggregates (arrays, Vec, structs, enums, tuples) are Python objects with reference identity.  Reading one in a value
  context (initialiser, argument, operand) copies it -- Rust's Copy / move -- while `&x`, `&mut x`, method receivers and
  index / field bases use the object itself.  References to scalars are Place objects (VarPlace / ElemPlace / ...).
"""
import ctypes
import math
import struct as _struct
from fractions import Fraction
from pathlib import Path

import numpy as np

from . import parser as P

F32 = np.float32
_libm = ctypes.CDLL('libm.so.6')
for _n in ('sinf', 'cosf', 'tanf', 'expf', 'exp2f', 'logf', 'log2f', 'log10f', 'atanf', 'asinf', 'acosf', 'sinhf', 'coshf', 'tanhf'):
    getattr(_libm, _n).restype = ctypes.c_float
    getattr(_libm, _n).argtypes = [ctypes.c_float]
for _n in ('powf', 'atan2f', 'fmodf', 'hypotf'):
    getattr(_libm, _n).restype = ctypes.c_float
    getattr(_libm, _n).argtypes = [ctypes.c_float, ctypes.c_float]
for _n in ('exp2', 'log2', 'cbrt'):
    getattr(_libm, _n).restype = ctypes.c_double
    getattr(_libm, _n).argtypes = [ctypes.c_double]

INT_BITS = {'u8': 8, 'u16': 16, 'u32': 32, 'u64': 64, 'u128': 128, 'usize': 64, 'i8': 8, 'i16': 16, 'i32': 32, 'i64': 64, 'i128': 128,
            'isize': 64, 'char': 32}
FLOAT_TYPES = ('f32', 'f64')


class RustPanic(Exception):
    pass


class InterpError(Exception):
    pass


class BreakEx(Exception):
    def __init__(self, label, value):
        self.label, self.value = label, value


class ContinueEx(Exception):
    def __init__(self, label):
        self.label = label


class ReturnEx(Exception):
    def __init__(self, value):
        self.value = value


# --------------------------------------------------------------------------------------------- values

class Int:
    __slots__ = ('v', 't')

    def __init__(self, v, t=None):
        self.v, self.t = v, t

    def __repr__(self):
        return '%d%s' % (self.v, self.t or '')

    def __index__(self):
        return self.v

    def __eq__(self, o):
        return isinstance(o, Int) and o.v == self.v

    def __hash__(self):
        return hash(self.v)


def wrap_int(v, t):
    bits = INT_BITS[t]
    if t[0] == 'u' or t == 'char':
        return v & ((1 << bits) - 1)
    m = 1 << (bits - 1)
    return ((v + m) & ((1 << bits) - 1)) - m


def in_range(v, t):
    return wrap_int(v, t) == v


_f32_cache = {}


def round_to_f32(fr):
    """Correctly rounded binary32 of an exact rational (round half to even), as Rust's literal parsing does."""
    d = float(fr)  # correctly rounded binary64
    if math.isinf(d) or d == 0.0:
        return F32(d)
    with np.errstate(over='ignore'):
        c = F32(d)
    cands = [c, np.nextafter(c, F32(np.inf)), np.nextafter(c, F32(-np.inf))]
    best, best_err = None, None
    for x in cands:
        if not np.isfinite(x):
            continue
        err = abs(Fraction(float(x)) - fr)
        if best is None or err < best_err or (err == best_err and (int(x.view(np.uint32)) & 1) == 0):
            best, best_err = x, err
    return best


def f32_from_text(text):
    r = _f32_cache.get(text)
    if r is None:
        r = round_to_f32(Fraction(text))
        _f32_cache[text] = r
    return r


class ULit:
    """An untyped float literal expression: evaluated once its type is known."""
    __slots__ = ('tree',)

    def __init__(self, tree):
        self.tree = tree

    def resolve(self, ty):
        return _ulit_eval(self.tree, ty)

    def __repr__(self):
        return 'ULit(%r)' % (self.tree,)


def _ulit_eval(t, ty):
    k = t[0]
    if k == 'lit':
        return f32_from_text(t[1]) if ty == 'f32' else float(t[1])
    if k == 'neg':
        return -_ulit_eval(t[1], ty)
    if k == 'val':  # an already typed operand cannot occur here
        raise InterpError('mixed literal tree')
    a, b = _ulit_eval(t[2], ty), _ulit_eval(t[3], ty)
    op = t[1]
    with np.errstate(all='ignore'):
        if op == '+':
            return a + b
        if op == '-':
            return a - b
        if op == '*':
            return a * b
        if op == '/':
            if ty == 'f64':
                return _fdiv(a, b)
            return a / b
        if op == '%':
            return math.fmod(a, b) if ty == 'f64' else F32(math.fmod(float(a), float(b)))
    raise InterpError('literal op ' + op)


def _fdiv(a, b):
    try:
        return a / b
    except ZeroDivisionError:
        if a == 0.0 or a != a:
            return math.nan
        return math.copysign(math.inf, a) * math.copysign(1.0, b)


class Arr:
    """[T; N], Vec<T>, Box<[T]>."""
    __slots__ = ('a', 'vec')

    def __init__(self, a, vec=False):
        self.a, self.vec = a, vec

    def __repr__(self):
        return 'Arr(%d)' % len(self.a)


class Slice:
    """&[T] / &mut [T]: a window into an Arr's storage.  `mut`: made by `&mut ..` (iterating it yields places)."""
    __slots__ = ('a', 'o', 'n', 'mut')

    def __init__(self, a, o, n, mut=False):
        self.a, self.o, self.n, self.mut = a, o, n, mut

    def __repr__(self):
        return 'Slice(%d..+%d)' % (self.o, self.n)


class Struct:
    __slots__ = ('name', 'f')

    def __init__(self, name, f):
        self.name, self.f = name, f

    def __repr__(self):
        return '%s%r' % (self.name, self.f)


class Enum:
    __slots__ = ('enum', 'variant', 'f')

    def __init__(self, enum, variant, f=None):
        self.enum, self.variant, self.f = enum, variant, f

    def __repr__(self):
        return '%s::%s%s' % (self.enum, self.variant, '' if self.f is None else repr(self.f))

    def __eq__(self, o):
        return isinstance(o, Enum) and o.enum == self.enum and o.variant == self.variant and values_equal(self.f, o.f)

    def __hash__(self):
        return hash((self.enum, self.variant))


class Range:
    __slots__ = ('lo', 'hi', 'incl')

    def __init__(self, lo, hi, incl):
        self.lo, self.hi, self.incl = lo, hi, incl

    def __repr__(self):
        return 'Range(%r, %r, %r)' % (self.lo, self.hi, self.incl)


class Uninit:
    """Default::default() of a type the interpreter cannot know; must be overwritten before it is read."""

    def __repr__(self):
        return '<default>'


UNINIT = Uninit()


class Place:
    __slots__ = ()


class VarPlace(Place):
    __slots__ = ('d', 'k')

    def __init__(self, d, k):
        self.d, self.k = d, k

    def get(self):
        return self.d[self.k]

    def set(self, v):
        self.d[self.k] = v


class ElemPlace(Place):
    __slots__ = ('a', 'i')

    def __init__(self, a, i):
        self.a, self.i = a, i

    def get(self):
        return self.a[self.i]

    def set(self, v):
        self.a[self.i] = v


FieldPlace = VarPlace  # a struct's field dict works like a scope


class TupleFieldPlace(Place):
    """`t.0 = v`: tuples are immutable Python tuples, so the write rebuilds the tuple in its own place."""
    __slots__ = ('parent', 'i')

    def __init__(self, parent, i):
        self.parent, self.i = parent, i

    def get(self):
        return deref(self.parent.get())[self.i]

    def set(self, v):
        t = list(deref(self.parent.get()))
        t[self.i] = v
        p = self.parent
        while isinstance(p.get(), Place):
            p = p.get()
        p.set(tuple(t))


class TempPlace(Place):
    __slots__ = ('v',)

    def __init__(self, v):
        self.v = v

    def get(self):
        return self.v

    def set(self, v):
        self.v = v


class ObjPlace(Place):
    """A reference to an aggregate (`&x`, `&mut x`): `*r = v` copies the contents into the object.  `mut`: made by
    `&mut ..` (iterating it yields places)."""
    __slots__ = ('o', 'mut')

    def __init__(self, o, mut=False):
        self.o, self.mut = o, mut

    def get(self):
        return self.o

    def set(self, v):
        o = self.o
        if isinstance(o, Arr):
            o.a[:] = list(seq_list(v))
        elif isinstance(o, Slice):
            src = list(seq_list(v))
            if len(src) != o.n:
                raise RustPanic('slice length mismatch in assignment')
            o.a[o.o:o.o + o.n] = src
        elif isinstance(o, Struct):
            o.f.clear()
            o.f.update(v.f)
            o.name = v.name
        elif isinstance(o, Enum):
            o.enum, o.variant, o.f = v.enum, v.variant, v.f
        else:
            raise InterpError('cannot assign through %r' % (o,))


class Closure:
    __slots__ = ('params', 'body', 'env')

    def __init__(self, params, body, env):
        self.params, self.body, self.env = params, body, env


class FnRef:
    __slots__ = ('item', 'self_type', 'gargs')

    def __init__(self, item, self_type=None, gargs=None):
        self.item, self.self_type, self.gargs = item, self_type, gargs


class Builtin:
    __slots__ = ('f', 'name')

    def __init__(self, f, name):
        self.f, self.name = f, name


class RIter:
    """A Rust iterator: a Python iterator, or a materialised list (double-ended / exact-size)."""
    __slots__ = ('it', 'lst', 'rem')

    def __init__(self, it=None, lst=None, rem=None):
        self.it, self.lst, self.rem = it, lst, rem

    def __iter__(self):
        if self.lst is not None:
            return iter(self.lst)
        return self.it

    def tolist(self):
        if self.lst is None:
            self.lst = list(self.it)
            self.it = None
        return self.lst


def is_agg(v):
    return isinstance(v, (Arr, Struct, Enum, tuple))


def copyval(v):
    """Rust's Copy / move of a value read from a place."""
    if isinstance(v, Arr):
        if v.vec:
            return v  # Vec / Box are moved, not copied
        return Arr([copyval(x) for x in v.a])
    if isinstance(v, Struct):
        return Struct(v.name, {k: copyval(x) for k, x in v.f.items()})
    if isinstance(v, Enum):
        if v.f is None:
            return v
        return Enum(v.enum, v.variant, {k: copyval(x) for k, x in v.f.items()})
    if isinstance(v, tuple):
        return tuple(copyval(x) for x in v)
    return v


def deepclone(v):
    if isinstance(v, Arr):
        return Arr([deepclone(x) for x in v.a], v.vec)
    if isinstance(v, Slice):
        return Arr([deepclone(x) for x in v.a[v.o:v.o + v.n]], True)
    if isinstance(v, Place):
        return deepclone(v.get())
    if isinstance(v, Struct):  # (a derived Clone clones the Vec / Box fields too; copyval would move them)
        return Struct(v.name, {k: deepclone(x) for k, x in v.f.items()})
    if isinstance(v, Enum) and v.f is not None:
        return Enum(v.enum, v.variant, {k: deepclone(x) for k, x in v.f.items()})
    if isinstance(v, tuple):
        return tuple(deepclone(x) for x in v)
    return copyval(v)


def deref(v):
    while isinstance(v, Place):
        v = v.get()
    return v


def seq_view(v):
    """(list, offset, length) of an array-like value."""
    v = deref(v)
    if isinstance(v, Arr):
        return v.a, 0, len(v.a)
    if isinstance(v, Slice):
        return v.a, v.o, v.n
    if isinstance(v, RIter):
        l = v.tolist()
        return l, 0, len(l)
    raise InterpError('not a sequence: %r' % (v,))


def seq_list(v):
    a, o, n = seq_view(v)
    return a[o:o + n] if (o or n != len(a)) else a


def values_equal(a, b):
    a, b = deref(a), deref(b)
    if isinstance(a, Int) and isinstance(b, Int):
        return a.v == b.v
    if isinstance(a, (Arr, Slice)) or isinstance(b, (Arr, Slice)):
        la, lb = seq_list(a), seq_list(b)
        return len(la) == len(lb) and all(values_equal(x, y) for x, y in zip(la, lb))
    if isinstance(a, Struct) and isinstance(b, Struct):
        return a.name == b.name and all(values_equal(a.f[k], b.f[k]) for k in a.f)
    if isinstance(a, dict) and isinstance(b, dict):
        return a.keys() == b.keys() and all(values_equal(a[k], b[k]) for k in a)
    if isinstance(a, tuple) and isinstance(b, tuple):
        return len(a) == len(b) and all(values_equal(x, y) for x, y in zip(a, b))
    if isinstance(a, ULit):
        a = a.resolve('f32' if isinstance(b, (np.floating, np.ndarray)) and not isinstance(b, float) else 'f64')
    if isinstance(b, ULit):
        b = b.resolve('f32' if isinstance(a, (np.floating, np.ndarray)) and not isinstance(a, float) else 'f64')
    r = a == b
    if isinstance(r, np.ndarray):
        return bool(r.all())
    return bool(r)


def is_f32(v):
    return isinstance(v, (np.float32, np.ndarray))


def float_type_of(v):
    if isinstance(v, float):
        return 'f64'
    if isinstance(v, (np.float32, np.ndarray)):
        return 'f32'
    return None


def truth(v):
    v = deref(v)
    if isinstance(v, (bool, np.bool_)):
        return bool(v)
    if isinstance(v, np.ndarray):
        if v.all():
            return True
        if not v.any():
            return False
        raise InterpError('a batched f32 comparison diverges between lanes: run this function with one lane')
    raise InterpError('not a bool: %r' % (v,))


NONE = Enum('Option', 'None')
UNIT = None


def some(v):
    return Enum('Option', 'Some', {'0': v})


def ok(v):
    return Enum('Result', 'Ok', {'0': v})


def err(v):
    return Enum('Result', 'Err', {'0': v})


class Env:
    __slots__ = ('scopes', 'self_type', 'generics', 'uses')

    def __init__(self, self_type=None, generics=None, uses=None):
        self.scopes = [{}]
        self.self_type = self_type
        self.generics = generics or {}
        self.uses = uses or {}

    def push(self):
        self.scopes.append({})

    def pop(self):
        self.scopes.pop()

    def lookup_scope(self, name):
        for s in reversed(self.scopes):
            if name in s:
                return s
        return None

    def bind(self, name, v):
        self.scopes[-1][name] = v


class Lazy:
    """A static / const: evaluated on first use."""
    __slots__ = ('item', 'interp', 'value', 'state', 'env', 'key')

    def __init__(self, item, interp, env=None, key=None):
        self.item, self.interp, self.value, self.state, self.env, self.key = item, interp, None, 0, env, key

    def get(self):
        if self.state == 2:
            return self.value
        if self.state == 1:
            raise InterpError('cyclic static ' + self.item[1])
        it = self.interp
        shared = it.shared_statics if self.key is not None and self.item[0] == 'static' else None
        if shared is not None and self.key in shared:  # an immutable `static` another interpreter built from the same source text
            self.value, self.state = shared[self.key], 2
            return self.value
        self.state = 1
        env = self.env if self.env is not None else Env(uses=self.item[5].uses if len(self.item) > 5 and self.item[5] is not None else None)
        v = it.ev(self.item[3], env)
        if self.item[2] is not None:
            v = it.coerce(v, self.item[2], env)
        self.value, self.state = v, 2
        if shared is not None:
            shared[self.key] = v
        return v


# --------------------------------------------------------------------------------------------- interpreter

class Interp:
    def __init__(self):
        self.globals = {}      # value namespace: name -> ('fn', ...) item | Lazy | struct/enum item
        self.types = {}        # type namespace: name -> struct / enum item
        self.impls = {}        # type name -> {method name -> fn item}
        self.shared_statics = None  # optional dict shared between interpreters: (sha1 of the source text, name) -> value of a `static`
        self.src_hash = {}     # file name -> sha1 of its text
        self.alt_methods = {}  # (type name, method name) -> [fn items] when an inherent and a trait method share a name
        self.macros = {}       # macro_rules
        self.trait_impls = {}  # type name -> [trait names it implements]
        self.traits = set()    # trait names
        self.native_methods = {}  # (type name, method) -> python callable(self_value, args)
        self.native_fns = {}   # path (last segment) -> python callable(*args): `extern "C"` functions bound by the harness
        self.supertraits = {}  # trait name -> [supertrait names]
        self.variant_of = {}   # enum variant name -> [enum name]  (for glob-imported variants)
        self.overflows = 0     # implicit integer wraps (a debug build would have panicked)
        self.trace_calls = None
        self.node_cache = {}
        self.files = []
        self.load_source((Path(__file__).parent / 'prelude.rs').read_text(), 'prelude.rs')

    # ---------------------------------------------------------------- loading
    def load_file(self, path):
        self.files.append(str(path))
        self.load_source(Path(path).read_text(), str(path))

    def load_source(self, src, fname):
        import hashlib
        self.src_hash[fname] = hashlib.sha1(src.encode()).hexdigest()
        self.register_items(P.parse_source(src, fname), fname)

    def register_items(self, items, fname, module=None):
        for it in items:
            k = it[0]
            if k == 'fn':
                self.globals[it[1]] = it
                if module:
                    self.globals[module + '::' + it[1]] = it
            elif k in ('const', 'static'):
                lz = Lazy(it, self, key=(self.src_hash[fname], it[1]) if fname in self.src_hash else None)
                self.globals[it[1]] = lz
                if module:
                    self.globals[module + '::' + it[1]] = lz
            elif k == 'struct':
                self.types[it[1]] = it
            elif k == 'enum':
                self.types[it[1]] = it
                for v in it[2]:
                    self.variant_of.setdefault(v[0], []).append(it[1])
            elif k == 'impl':
                tname = self.type_name(it[1])
                d = self.impls.setdefault(tname, {})
                if it[2] is not None:  # `impl Trait for Type`: the trait's default methods are Type's too
                    tr = self.type_name(it[2])
                    if tr not in self.trait_impls.setdefault(tname, []):
                        self.trait_impls[tname].append(tr)
                for sub in it[3]:
                    if sub[0] == 'fn':
                        old = d.get(sub[1])
                        if isinstance(old, tuple) and old[0] == 'fn' and old is not sub and len(old[3]) != len(sub[3]):
                            # an inherent method and a trait method of one name on one type (`Dsp::synth` and
                            # `<Dsp as SynthBackend>::synth`): rustc tells them apart by the path / the receiver's static type, which
                            # are not tracked here; the two that occur differ in their parameter count, and call_fn picks by that
                            self.alt_methods.setdefault((tname, sub[1]), [old]).append(sub)
                        d[sub[1]] = sub
                    elif sub[0] in ('const', 'static'):
                        d[sub[1]] = Lazy(sub, self, Env(self_type=tname, uses=sub[5].uses))
            elif k == 'mod':
                if any('cfg ( test )' in a for a in it[3]):
                    self.register_items(it[2], fname, 'tests')  # reachable as tests::name only
                    continue
                self.register_items(it[2], fname, it[1])
            elif k == 'macro_rules':
                self.macros[it[1]] = self.parse_macro_rules(it[2])
            elif k == 'macro_item':
                self.expand_item_macro(it, fname, module)
            elif k == 'unparsed':
                self.globals.setdefault('__unparsed__', []).append(it[1])
            elif k == 'trait':
                self.traits.add(it[1])
                for sub in it[2]:  # default methods
                    if sub[0] == 'fn' and sub[6] is not None:
                        self.impls.setdefault('<trait ' + it[1] + '>', {})[sub[1]] = sub

    def expand_item_macro(self, it, fname, module):
        name, toks = it[1], it[2]
        if name == 'lazy_static':
            self.register_items(P.parse_tokens_as_items(toks, fname), fname, module)
            return
        if name in self.macros:
            out = self.expand_macro(name, toks)
            self.register_items(P.parse_tokens_as_items(out, fname), fname, module)
            return
        # unknown item macros (e.g. bitflags!, support_audio_codec!) are not on the DSP path

    def type_name(self, ty):
        if ty is None:
            return None
        k = ty[0]
        if k == 'tpath':
            return ty[1][-1]
        if k == 'tref':
            return self.type_name(ty[2])
        if k == 'tarray' or k == 'tslice':
            return '[]'
        return k

    # ---------------------------------------------------------------- macro_rules (no repetitions)
    def parse_macro_rules(self, toks):
        arms, p = [], P.Parser(list(toks) + [P.Tok('eof', '', 0)])
        while p.cur.k != 'eof':
            _, pat = p.token_tree()
            p.expect('=>')
            _, body = p.token_tree()
            p.eat(';')
            arms.append((pat, body))
        return arms

    def expand_macro(self, name, toks):
        for pat, body in self.macros[name]:
            b = self.match_macro(pat, toks)
            if b is not None:
                out, i = [], 0
                while i < len(body):
                    t = body[i]
                    if t.k == 'punct' and t.s == '$' and i + 1 < len(body) and body[i + 1].s in b:
                        out.extend(b[body[i + 1].s])
                        i += 2
                    else:
                        out.append(t)
                        i += 1
                return out
        raise InterpError('no macro_rules arm of %s! matches' % name)

    @staticmethod
    def match_macro(pat, toks):
        # pattern: ($a:frag SEP $b:frag ...) with single-token separators
        binds, i, j = {}, 0, 0
        while i < len(pat):
            t = pat[i]
            if t.s == '$' and i + 3 < len(pat) + 1 and pat[i + 2].s == ':':
                var = pat[i + 1].s
                sep = pat[i + 4].s if i + 4 < len(pat) else None
                depth, start = 0, j
                while j < len(toks):
                    s = toks[j]
                    if s.k == 'punct':
                        if s.s in '([{' and len(s.s) == 1:
                            depth += 1
                        elif s.s in ')]}' and len(s.s) == 1:
                            depth -= 1
                        elif depth == 0 and sep is not None and s.s == sep:
                            break
                    j += 1
                if j == start:
                    return None
                frag = toks[start:j]
                # keep an expression fragment as one operand
                binds[var] = [P.Tok('punct', '(', 0)] + frag + [P.Tok('punct', ')', 0)] if pat[i + 3].s == 'expr' and len(frag) > 1 else frag
                i += 4
            else:
                if j >= len(toks) or toks[j].s != t.s:
                    return None
                i += 1
                j += 1
        return binds if j == len(toks) else None

    # ---------------------------------------------------------------- public helpers
    def call(self, name, *args, generics=None):
        """Call a loaded function by (possibly qualified) name with interpreter values.  Aggregates passed for
        reference parameters are passed by reference (the callee's writes are visible in the caller's object)."""
        f = self.resolve_value([s for s in name.split('::')], Env(), None)
        if isinstance(f, FnRef):
            args = self.by_ref_args(f.item, args)
        return self.call_value(f, list(args), generics)

    def call_method(self, type_name, name, self_val, *args, generics=None):
        item = self.impls[type_name][name]
        sv = ObjPlace(self_val) if isinstance(self_val, (Struct, Enum, Arr)) and item[4] != 'value' else self_val
        return self.call_fn(item, self.by_ref_args(item, args), generics, type_name, None, self_val=sv)

    def exec_where(self, item, key, bindings, self_type=None):
        """Run, in order, the top-level statements of a function's body whose syntax tree mentions `key` (an identifier),
        in an environment holding `bindings`.  For code that is written inline in a long method (Vorbis inverse coupling,
        the floor-1 neighbour precomputation, ...): the statements are the reference's, only the surroundings are skipped."""
        blk = self.node_cache.get(id(item))
        if blk is None:
            blk = item[8].parse_body(item[6])
            self.node_cache[id(item)] = blk
        env = Env(self_type=self_type, uses=item[8].uses)
        for k, v in bindings.items():
            env.bind(k, v)
        ran = 0
        keys = () if callable(key) else (key,) if isinstance(key, str) else tuple(key)
        for st in blk[1]:
            text = repr(st)
            if key(text) if callable(key) else all((("'%s'" % k) in text) if not k.startswith('!') else (("'%s'" % k[1:]) not in text) for k in keys):
                ran += 1
                if st[0] == 'expr':
                    self.ev(st[1], env)
                elif st[0] == 'let':
                    v = self.ev(st[3], env)
                    if st[2] is not None:
                        v = self.coerce(v, st[2], env)
                    self.match(st[1], v, env, True)
        if not ran:
            raise InterpError('no statement mentions %s' % key)
        return env

    def body_of(self, item):
        blk = self.node_cache.get(id(item))
        if blk is None:
            blk = item[8].parse_body(item[6])
            self.node_cache[id(item)] = blk
        return blk

    def find_stmts(self, item, pred):
        """Statements at ANY depth of a function's body whose syntax tree satisfies pred(repr), innermost match first."""
        found = []

        def walk(node):
            if isinstance(node, tuple):
                if node and node[0] == 'block':
                    for st in node[1]:
                        inner_before = len(found)
                        walk(st)
                        if len(found) == inner_before and pred(repr(st)):
                            found.append(st)
                    if node[2] is not None:  # the block's tail expression counts as its last statement
                        st = ('expr', node[2])
                        inner_before = len(found)
                        walk(node[2])
                        if len(found) == inner_before and pred(repr(st)):
                            found.append(st)
                    return
                for x in node:
                    walk(x)
            elif isinstance(node, list):
                for x in node:
                    walk(x)
        walk(self.body_of(item))
        return found

    def local_fn(self, item, name):
        """A fn item nested (at any depth) in another function's body."""
        st = self.find_stmts(item, lambda t: t.startswith("('item', ('fn', '%s'" % name))
        if not st:
            raise InterpError('no nested fn %s' % name)
        return FnRef(st[0][1], None, None)

    def exec_stmt(self, item, st, bindings, self_type=None):
        env = Env(self_type=self_type, uses=item[8].uses)
        for k, v in bindings.items():
            env.bind(k, v)
        if st[0] == 'expr':
            return self.ev(st[1], env), env
        if st[0] == 'let':
            v = self.ev(st[3], env)
            if st[2] is not None:
                v = self.coerce(v, st[2], env)
            self.match(st[1], v, env, True)
            return v, env
        raise InterpError('cannot execute %s' % st[0])

    @staticmethod
    def by_ref_args(item, args):
        out = []
        for (pat, ty), a in zip(item[3], args):
            if ty is not None and ty[0] == 'tref' and isinstance(a, (Arr, Struct)):
                a = ObjPlace(a, bool(ty[1]))
            out.append(a)
        return out + list(args[len(item[3]):])

    def new_struct(self, name, **fields):
        return Struct(name, dict(fields))

    # ---------------------------------------------------------------- coercion to declared types
    def coerce(self, v, ty, env):
        if ty is None:
            return v
        k = ty[0]
        if k == 'tpath':
            name = ty[1][-1]
            if name in env.generics and not isinstance(env.generics[name], Int):
                g = env.generics[name]
                if isinstance(g, tuple):
                    return self.coerce(v, g, env)
                return v
            if name in FLOAT_TYPES:
                if isinstance(v, ULit):
                    return v.resolve(name)
                if isinstance(v, Int) and v.t is None:
                    raise InterpError('integer literal where %s is expected' % name)
                return v
            if name in INT_BITS:
                if isinstance(v, Int) and v.t is None:
                    if not in_range(v.v, name):
                        raise InterpError('literal %d out of range for %s' % (v.v, name))
                    return Int(v.v, name)
                return v
            if name in ('Vec', 'Box', 'Option') and ty[2]:
                inner = ty[2][0][1] if ty[2][0][0] == 'gtype' else None
                if inner is None:
                    return v
                if name == 'Option':
                    if isinstance(v, Enum) and v.variant == 'Some':
                        v.f['0'] = self.coerce(v.f['0'], inner, env)
                    return v
                if name == 'Vec':
                    return self.coerce_seq(v, inner, env)
                return self.coerce(v, inner, env)
            if name == 'Wrapping' and ty[2] and isinstance(v, Struct):
                v.f['0'] = self.coerce(v.f['0'], ty[2][0][1], env)
                return v
            if isinstance(v, Struct) and ty[2]:
                sd = self.types.get(v.name)
                if sd is not None and sd[0] == 'struct' and len(sd) > 5:
                    gnames = [g[1] for g in sd[5] if g[0] == 'type']
                    gmap = {n: a[1] for n, a in zip(gnames, [a for a in ty[2] if a[0] == 'gtype'])}
                    for fname, fty in sd[3]:
                        if fty[0] == 'tpath' and fty[1][-1] in gmap and fname in v.f:
                            v.f[fname] = self.coerce(v.f[fname], gmap[fty[1][-1]], env)
            return v
        if k == 'tref':
            return self.coerce(v, ty[2], env)
        if k == 'tarray':
            v2 = self.coerce_seq(v, ty[1], env)
            if isinstance(deref(v2), (Arr, Slice)):
                want = self.ev(ty[2], env)
                if isinstance(want, Int):
                    n = seq_view(v2)[2]
                    if n != want.v:
                        raise RustPanic('expected an array of %d elements, found %d' % (want.v, n))
            return v2
        if k == 'tslice':
            return self.coerce_seq(v, ty[1], env)
        if k == 'ttuple' and isinstance(v, tuple):
            return tuple(self.coerce(x, t, env) for x, t in zip(v, ty[1]))
        return v

    def coerce_seq(self, v, elem_ty, env):
        d = deref(v)
        if not isinstance(d, (Arr, Slice)):
            return v
        a, o, n = seq_view(d)
        if n == 0:
            return v
        first = a[o]
        if (isinstance(first, Int) and first.t is None) or isinstance(first, ULit) or is_agg(first):
            if is_agg(first) and elem_ty[0] == 'tpath' and not elem_ty[2] and elem_ty[1][-1] not in env.generics:
                return v
            for i in range(o, o + n):
                a[i] = self.coerce(a[i], elem_ty, env)
        return v

    def adopt_type(self, v, like):
        """Give an untyped literal the type of the value it replaces / is combined with."""
        if isinstance(v, ULit):
            ft = float_type_of(deref(like))
            return v.resolve(ft) if ft else v
        if isinstance(v, Int) and v.t is None:
            l = deref(like)
            if isinstance(l, Int) and l.t is not None:
                return Int(v.v, l.t)
        return v

    # ---------------------------------------------------------------- name resolution
    def resolve_value(self, segs, env, gargs):
        name = segs[-1]
        if len(segs) >= 2 and segs[0] in env.uses and env.lookup_scope(segs[0]) is None:
            segs = env.uses[segs[0]] + segs[1:]
        if len(segs) == 1:
            sc = env.lookup_scope(name)
            if sc is not None:
                return sc[name]
            if name in env.generics:
                return env.generics[name]
            g = self.globals.get(name)
            if g is not None:
                return self.global_value(g, gargs)
            if name == 'Self' and env.self_type:
                return ('type', env.self_type)
            b = self.builtin_fn(segs)
            if b is not None:
                return b
            if name in self.variant_of and len(self.variant_of[name]) == 1:
                return self.enum_variant(self.variant_of[name][0], name)
            if name in self.types and self.types[name][0] == 'struct':
                if self.types[name][2] == 'unit':  # `struct Marker;` used as a value
                    return Struct(name, {})
                return ('ctor', name)
            raise InterpError('unresolved name %s' % name)
        # qualified
        q = '::'.join(segs[-2:])
        g = self.globals.get(q)
        if g is not None:
            return self.global_value(g, gargs)
        tname = segs[-2]
        if tname == 'Self' and env.self_type:
            tname = env.self_type
        gt = env.generics.get(tname) if getattr(env, 'generics', None) else None
        if isinstance(gt, tuple) and gt[0] == 'tpath':  # `C::new(..)` with C a type parameter the caller named (turbofish)
            tname = gt[1][-1]
        if tname in self.types and self.types[tname][0] == 'enum':
            for v in self.types[tname][2]:
                if v[0] == name:
                    return self.enum_variant(tname, name)
        if tname in self.impls and name in self.impls[tname]:
            m = self.impls[tname][name]
            if isinstance(m, Lazy):
                return m.get()
            return FnRef(m, tname, gargs)
        if tname in self.traits:  # `Trait::method(receiver, ..)`: dispatch on the receiver's type
            def via_trait(recv, *rest, _n=name, _t=tname):
                base = deref(recv)
                from . import stdext
                while stdext.is_std(base) and isinstance(base, stdext.Cell):
                    base = base.v
                ty = base.name if isinstance(base, Struct) else (base.enum if isinstance(base, Enum) else None)
                m = self.impls.get(ty, {}).get(_n) or self.trait_default(ty, _n) or self.impls.get('<trait ' + _t + '>', {}).get(_n)
                if m is None:
                    raise InterpError('no %s::%s for %r' % (_t, _n, ty))
                self_val = copyval(base) if m[4] == 'value' else ObjPlace(base)
                return self.call_fn(m, list(rest), None, ty, None, self_val=self_val)
            return Builtin(via_trait, tname + '::' + name)
        b = self.builtin_fn(segs)
        if b is not None:
            return b
        g = self.globals.get(name)
        if g is not None and tname not in self.types:
            return self.global_value(g, gargs)  # module-qualified: crate::a::b::f, super::f, self::f
        if name == 'default' and tname not in self.types and len(tname) <= 2 and tname.isupper():
            return Builtin(lambda: UNINIT, 'T::default')  # of a type parameter: a placeholder until something typed overwrites it
        raise InterpError('unresolved path %s' % '::'.join(segs))

    def global_value(self, g, gargs):
        if isinstance(g, Lazy):
            return g.get()
        if isinstance(g, Builtin):  # an `extern "C"` function bound by the harness (rsinterp/ffi.py)
            return g
        if g[0] == 'fn':
            return FnRef(g, None, gargs)
        return g

    def enum_variant(self, ename, vname):
        for v in self.types[ename][2]:
            if v[0] == vname:
                if v[1] == 'unit':
                    return Enum(ename, vname)
                return ('variant_ctor', ename, vname)
        raise InterpError('no variant %s::%s' % (ename, vname))

    # ---------------------------------------------------------------- calls
    def call_value(self, f, args, gargs=None, env=None):
        if isinstance(f, FnRef):
            return self.call_fn(f.item, args, f.gargs if f.gargs else gargs, f.self_type, env)
        if isinstance(f, Closure):
            e = f.env
            e.push()
            try:
                for (pat, ty), a in zip(f.params, args):
                    if ty is not None:
                        a = self.coerce(a, ty, e)
                    if not self.match(pat, a, e, True):
                        raise RustPanic('closure argument pattern mismatch')
                try:
                    return self.ev(f.body, e)
                except ReturnEx as r:
                    return r.value
            finally:
                e.pop()
        if isinstance(f, Builtin):
            return f.f(*args)
        if isinstance(f, tuple):
            if f[0] == 'ctor':
                sd = self.types[f[1]]
                return Struct(f[1], {str(i): a for i, a in enumerate(args)})
            if f[0] == 'variant_ctor':
                return Enum(f[1], f[2], {str(i): a for i, a in enumerate(args)})
        raise InterpError('not callable: %r' % (f,))

    def call_fn(self, item, args, gargs=None, self_type=None, caller_env=None, self_val=None):
        _, name, gen, params, self_kind, ret, body, attrs, parser = item
        env = Env(self_type=self_type, uses=parser.uses)
        if gen:
            cg = [g for g in gen if g[0] != 'lifetime']
            if gargs:
                for (kind, gname), ga in zip(cg, gargs):
                    if ga[0] == 'gconst':
                        env.generics[gname] = self.ev(ga[1], caller_env or Env())
                    else:
                        t = ga[1]
                        if kind == 'const':  # a const argument written as a path (e.g. `N` of the enclosing fn)
                            env.generics[gname] = self.ev(('path', t[1], None), caller_env or Env()) if t[0] == 'tpath' else None
                        else:
                            if t[0] == 'tpath' and caller_env is not None and t[1][-1] in caller_env.generics:
                                t = caller_env.generics[t[1][-1]]
                            env.generics[gname] = t
        if self.trace_calls is not None:
            self.trace_calls.append(name)
        if self_kind is not None:
            if self_val is None:
                self_val, args = args[0], args[1:]
            env.bind('self', self_val)
        if len(args) != len(params):
            for alt in self.alt_methods.get((self_type, name), ()):
                if alt is not item and len(alt[3]) == len(args):
                    return self.call_fn(alt, args, gargs, self_type, caller_env, self_val=self_val)
            raise InterpError('%s expects %d arguments, got %d' % (name, len(params), len(args)))
        for (pat, ty), a in zip(params, args):
            a = self.coerce(a, ty, env)
            if not self.match(pat, a, env, True):
                raise RustPanic('irrefutable pattern failed in ' + name)
        if body is None:
            raise InterpError('function %s has no body' % name)
        blk = self.node_cache.get(id(item))
        if blk is None:
            blk = parser.parse_body(body)
            self.node_cache[id(item)] = blk
        try:
            v = self.ev_block(blk, env, new_scope=False)
        except ReturnEx as r:
            v = r.value
        if ret is not None:
            v = self.coerce(v, ret, env)
        return v

    # ---------------------------------------------------------------- patterns
    def match(self, pat, v, env, bind):
        k = pat[0]
        if k == 'pident':
            name = pat[1]
            # a constant or unit variant in scope makes this a path pattern
            if name[0].isupper():
                g = self.globals.get(name)
                if g is not None and not (isinstance(g, tuple) and g[0] == 'fn'):
                    return values_equal(self.global_value(g, None), v)
                if name in self.variant_of and env.lookup_scope(name) is None:
                    dv = deref(v)
                    return isinstance(dv, Enum) and dv.variant == name
            if bind:
                env.bind(name, v)
            return True
        if k == 'pwild' or k == 'prest':
            return True
        if k == 'pbind':
            if pat[2] is not None and not self.match(pat[2], v, env, bind):
                return False
            if bind:
                env.bind(pat[1], v)
            return True
        if k == 'plit':
            lit = self.ev(pat[1], env)
            return values_equal(lit, v)
        if k == 'prange':
            dv = deref(v)
            lo = self.ev(pat[1], env) if pat[1] is not None else None
            hi = self.ev(pat[2], env) if pat[2] is not None else None
            x = dv.v if isinstance(dv, Int) else dv
            if lo is not None and x < (lo.v if isinstance(lo, Int) else lo):
                return False
            if hi is not None:
                h = hi.v if isinstance(hi, Int) else hi
                return x <= h if pat[3] else x < h
            return True
        if k == 'ptuple':
            dv = deref(v)
            if dv is UNIT and not pat[1]:
                return True  # `()` against the unit value (`Ok(()) => ..`)
            if not isinstance(dv, tuple):
                raise InterpError('tuple pattern against %r' % (dv,))
            pats = pat[1]
            if any(p[0] == 'prest' for p in pats):
                i = [p[0] for p in pats].index('prest')
                head, tail = pats[:i], pats[i + 1:]
                return all(self.match(p, x, env, bind) for p, x in zip(head, dv)) and \
                    all(self.match(p, x, env, bind) for p, x in zip(tail, dv[len(dv) - len(tail):]))
            if len(pats) != len(dv):
                raise InterpError('tuple pattern arity')
            return all(self.match(p, x, env, bind) for p, x in zip(pats, dv))
        if k == 'pref':
            inner = deref_once(v)
            return self.match(pat[1], copyval(inner) if is_agg(inner) else inner, env, bind)
        if k == 'por':
            return any(self.match(p, v, env, bind) for p in pat[1])
        if k == 'ppath':
            target = self.resolve_value(pat[1], env, None)
            return values_equal(target, v)
        if k == 'ptstruct':
            dv = deref(v)
            path = pat[1]
            vname = path[-1]
            if isinstance(dv, Enum):
                if dv.variant != vname:
                    return False
                fields = [dv.f[str(i)] for i in range(len(dv.f))] if dv.f else []
            elif isinstance(dv, Struct):
                if dv.name != vname:
                    return False
                fields = [dv.f[str(i)] for i in range(len(dv.f))]
            else:
                return False
            pats = pat[2]
            if any(p[0] == 'prest' for p in pats):
                pats = [p for p in pats if p[0] != 'prest']
            return all(self.match(p, x, env, bind) for p, x in zip(pats, fields))
        if k == 'pstruct':
            dv = deref(v)
            vname = pat[1][-1]
            if isinstance(dv, Enum):
                if dv.variant != vname:
                    return False
            elif isinstance(dv, Struct):
                if dv.name != vname and vname != 'Self':
                    return False
            else:
                return False
            for fname, fp in pat[2]:
                if not self.match(fp, dv.f[fname], env, bind):
                    return False
            return True
        if k == 'pslice':
            lst = seq_list(v)
            pats = pat[1]
            if any(p[0] == 'prest' for p in pats):
                i = [p[0] for p in pats].index('prest')
                head, tail = pats[:i], pats[i + 1:]
                if len(lst) < len(head) + len(tail):
                    return False
                return all(self.match(p, x, env, bind) for p, x in zip(head, lst)) and \
                    all(self.match(p, x, env, bind) for p, x in zip(tail, lst[len(lst) - len(tail):]))
            return len(pats) == len(lst) and all(self.match(p, x, env, bind) for p, x in zip(pats, lst))
        raise InterpError('pattern ' + k)

    # ---------------------------------------------------------------- blocks and statements
    def ev_block(self, blk, env, new_scope=True):
        _, stmts, tail = blk
        if new_scope:
            env.push()
        try:
            for st in stmts:  # items are visible in the whole block
                if st[0] == 'item':
                    self.local_item(st[1], env)
            for st in stmts:
                k = st[0]
                if k == 'expr':
                    self.ev(st[1], env)
                elif k == 'let':
                    _, pat, ty, init, els = st
                    if init is None:
                        if pat[0] in ('pident', 'pbind'):
                            env.bind(pat[1], UNINIT)
                        continue
                    v = self.ev(init, env)
                    if v is UNINIT and ty is not None:  # `let x: T = Default::default();`
                        from . import stdext
                        v = stdext.default_of(self, ty, env)
                    if ty is not None:
                        v = self.coerce(v, ty, env)
                    if not self.match(pat, v, env, True):
                        if els is None:
                            raise RustPanic('refutable pattern in let')
                        self.ev(els, env)
            if tail is not None:
                return self.ev(tail, env)
            return UNIT
        finally:
            if new_scope:
                env.pop()

    def local_item(self, it, env):
        k = it[0]
        if k == 'fn':
            env.bind(it[1], FnRef(it, env.self_type, None))
        elif k in ('const', 'static'):
            lz = Lazy(it, self, env)
            env.bind(it[1], LazyLocal(lz))
        elif k in ('struct', 'enum', 'impl', 'macro_rules'):
            self.register_items([it], '<local>')

    # ---------------------------------------------------------------- expressions
    def ev(self, e, env):
        """Value context: aggregates read from places are copied."""
        k = e[0]
        if k == 'path' or k == 'index' or k == 'field':
            v = self.evr(e, env)
            if isinstance(v, (Arr, Struct, tuple)) or (isinstance(v, Enum) and v.f is not None):
                return copyval(v)
            return v
        return getattr(self, 'e_' + k)(e, env)

    def evr(self, e, env):
        """Reference context: the object itself (no copy); scalars by value."""
        k = e[0]
        if k == 'path':
            segs = e[1]
            if len(segs) == 1:
                name = segs[0]
                for s in reversed(env.scopes):
                    if name in s:
                        v = s[name]
                        if type(v) is LazyLocal:
                            return v.lz.get()
                        return v
            v = self.resolve_value(segs, env, e[2])
            if isinstance(v, FnRef) and e[2]:
                v = FnRef(v.item, v.self_type, e[2])
            return v
        if k == 'index':
            base = deref(self.evr(e[1], env))
            idx = self.ev(e[2], env)
            return self.index_value(base, idx)
        if k == 'field':
            base = deref(self.evr(e[1], env))
            return self.field_value(base, e[2])
        if k == 'paren':
            return self.evr(e[1], env)
        if k == 'deref':
            v = self.evr(e[1], env)
            if isinstance(v, Place):
                return v.get()
            return v
        return getattr(self, 'e_' + k)(e, env)

    def index_value(self, base, idx):
        idx = deref(idx)
        if isinstance(idx, Int):
            if isinstance(base, Arr):
                try:
                    if idx.v < 0:
                        raise IndexError
                    return base.a[idx.v]
                except IndexError:
                    raise RustPanic('index out of bounds: the len is %d but the index is %d' % (len(base.a), idx.v))
            if isinstance(base, Slice):
                if not 0 <= idx.v < base.n:
                    raise RustPanic('index out of bounds: the len is %d but the index is %d' % (base.n, idx.v))
                return base.a[base.o + idx.v]
            raise InterpError('cannot index %r' % (base,))
        if isinstance(idx, Range):
            a, o, n = seq_view(base)
            lo = idx.lo.v if idx.lo is not None else 0
            hi = (idx.hi.v + (1 if idx.incl else 0)) if idx.hi is not None else n
            if lo > hi:
                raise RustPanic('slice index starts at %d but ends at %d' % (lo, hi))
            if hi > n:
                raise RustPanic('range end index %d out of range for slice of length %d' % (hi, n))
            return Slice(a, o + lo, hi - lo)
        raise InterpError('index with %r' % (idx,))

    def field_value(self, base, name):
        if isinstance(base, Struct):
            try:
                return base.f[name]
            except KeyError:
                raise InterpError('no field %s on %s' % (name, base.name))
        if isinstance(base, tuple):
            return base[int(name)]
        if isinstance(base, Enum) and base.f is not None and name in base.f:
            return base.f[name]
        raise InterpError('field %s of %r' % (name, base))

    def place(self, e, env):
        k = e[0]
        if k == 'path' and len(e[1]) == 1:
            sc = env.lookup_scope(e[1][0])
            if sc is not None:
                v = sc[e[1][0]]
                if isinstance(v, Place):  # a &mut binding used as a place: assignment rebinds the variable
                    return VarPlace(sc, e[1][0])
                return VarPlace(sc, e[1][0])
            return TempPlace(self.evr(e, env))
        if k == 'index':
            base = deref(self.evr(e[1], env))
            idx = deref(self.ev(e[2], env))
            if isinstance(idx, Int):
                if isinstance(base, Arr):
                    if not 0 <= idx.v < len(base.a):
                        raise RustPanic('index out of bounds: the len is %d but the index is %d' % (len(base.a), idx.v))
                    return ElemPlace(base.a, idx.v)
                if isinstance(base, Slice):
                    if not 0 <= idx.v < base.n:
                        raise RustPanic('index out of bounds: the len is %d but the index is %d' % (base.n, idx.v))
                    return ElemPlace(base.a, base.o + idx.v)
            return ObjPlace(self.index_value(base, idx))
        if k == 'field':
            base = deref(self.evr(e[1], env))
            if isinstance(base, Struct):
                return FieldPlace(base.f, e[2])
            if isinstance(base, Enum):
                return FieldPlace(base.f, e[2])
            if isinstance(base, tuple):
                return TupleFieldPlace(self.place(e[1], env), int(e[2]))
            raise InterpError('field place on %r' % (base,))
        if k == 'deref':
            v = self.evr(e[1], env)
            if isinstance(v, Place):
                return v
            if isinstance(v, (Arr, Slice, Struct, Enum)):
                return ObjPlace(v)
            return TempPlace(v)
        if k == 'paren':
            return self.place(e[1], env)
        return TempPlace(self.evr(e, env))

    # literals
    def e_int(self, e, env):
        return Int(e[1], e[2])

    def e_float(self, e, env):
        if e[2] == 'f32':
            return f32_from_text(e[1])
        if e[2] == 'f64':
            return float(e[1])
        return ULit(('lit', e[1]))

    def e_bool(self, e, env):
        return e[1]

    def e_str(self, e, env):
        s = e[1]
        if s.startswith('b"'):  # a byte-string literal is a `&[u8; N]`
            raw = s[2:-1].encode('ascii').decode('unicode_escape').encode('latin-1')
            return Arr([Int(b, 'u8') for b in raw], False)
        return s

    def e_char(self, e, env):
        return Int(e[1], 'u8' if e[2] else 'char')

    def e_paren(self, e, env):
        return self.ev(e[1], env)

    def e_tuple(self, e, env):
        if not e[1]:
            return UNIT
        return tuple(self.ev(x, env) for x in e[1])

    def e_array(self, e, env):
        return Arr([self.ev(x, env) for x in e[1]])

    def e_repeat(self, e, env):
        n = deref(self.ev(e[2], env))
        v = self.ev(e[1], env)
        if is_agg(v):
            return Arr([copyval(v) for _ in range(n.v)])
        return Arr([v] * n.v)

    def e_struct(self, e, env):
        segs = e[1]
        name = segs[-1]
        if name == 'Self':
            name = env.self_type
        fields = {fn: self.ev(fe, env) for fn, fe in e[2]}
        if len(segs) >= 2 and segs[-2] in self.types and self.types[segs[-2]][0] == 'enum':
            return Enum(segs[-2], name, fields)
        if name in self.variant_of and name not in self.types:
            return Enum(self.variant_of[name][0], name, fields)
        if e[3] is not None:
            base = deref(self.ev(e[3], env))
            if base is UNINIT:  # `..Default::default()`
                from . import stdext
                base = stdext.default_of(self, ('tpath', [name], []))
            for k2, v2 in base.f.items():
                fields.setdefault(k2, v2)
        sd = self.types.get(name)
        if sd is not None and sd[0] == 'struct':
            for fname, fty in sd[3]:
                if fname in fields:
                    if fields[fname] is UNINIT:  # `field: Default::default()`: the declared type says of what
                        from . import stdext
                        d = stdext.default_of(self, fty, env)
                        if d is not UNINIT:
                            fields[fname] = d
                            continue
                    fields[fname] = self.coerce(fields[fname], fty, env)
        return Struct(name, fields)

    def e_range(self, e, env):
        lo = deref(self.ev(e[1], env)) if e[1] is not None else None
        hi = deref(self.ev(e[2], env)) if e[2] is not None else None
        if isinstance(lo, Int) and isinstance(hi, Int):
            if lo.t is None and hi.t is not None:
                lo = Int(lo.v, hi.t)
            elif hi.t is None and lo.t is not None:
                hi = Int(hi.v, lo.t)
        return Range(lo, hi, e[3])

    def e_ref(self, e, env):
        inner = e[2]
        v = self.evr(inner, env)
        if isinstance(v, (Arr, Struct, tuple)) or (isinstance(v, Enum) and v.f is not None):
            return ObjPlace(v, e[1])  # a reference to an aggregate: never copied when passed on, assignment through it writes the object
        if isinstance(v, Slice):
            return Slice(v.a, v.o, v.n, True) if e[1] and not v.mut else v
        if isinstance(v, ObjPlace) and e[1] and not v.mut:
            return ObjPlace(v.o, True)
        if isinstance(v, (Enum, Place, RIter, Closure, FnRef, str)):
            return v
        if e[1]:  # &mut scalar
            return self.place(inner, env)
        return v

    def e_deref(self, e, env):
        v = self.evr(e[1], env)
        if isinstance(v, Place):
            v = v.get()
        if is_agg(v):
            return copyval(v)
        return v

    def e_unary(self, e, env):
        v = deref(self.ev(e[2], env))
        if e[1] == '-':
            if isinstance(v, Int):
                if v.t is None:
                    return Int(-v.v)
                r = wrap_int(-v.v, v.t)
                if r != -v.v:
                    self.overflows += 1
                return Int(r, v.t)
            if isinstance(v, ULit):
                return ULit(('neg', v.tree))
            if isinstance(v, Struct):
                return self.op_trait(v, 'neg', [])
            return -v
        # '!'
        if isinstance(v, (bool, np.bool_)):
            return not v
        if isinstance(v, np.ndarray):
            return ~v
        if isinstance(v, Int):
            if v.t is None:
                return Int(~v.v)
            return Int(wrap_int(~v.v, v.t), v.t)
        raise InterpError('! on %r' % (v,))

    def e_and(self, e, env):
        return truth(self.ev(e[1], env)) and truth(self.ev(e[2], env))

    def e_or(self, e, env):
        return truth(self.ev(e[1], env)) or truth(self.ev(e[2], env))

    def e_binary(self, e, env):
        a = self.ev(e[2], env)
        b = self.ev(e[3], env)
        return self.binop(e[1], a, b)

    def op_trait(self, a, method, args):
        if a.name == 'Wrapping':
            return self.wrapping_op(a, method, args)
        m = self.impls.get(a.name, {}).get(method)
        if m is None:
            raise InterpError('no impl of %s for %s' % (method, a.name))
        return self.call_fn(m, [a] + args, None, a.name)

    def wrapping_op(self, a, method, args):
        x = a.f['0']
        if method == 'neg':
            return Struct('Wrapping', {'0': Int(wrap_int(-x.v, x.t), x.t)})
        y = args[0].f['0'] if isinstance(args[0], Struct) else args[0]
        t = x.t or y.t or 'i32'
        op = {'add': lambda p, q: p + q, 'sub': lambda p, q: p - q, 'mul': lambda p, q: p * q, 'shl': lambda p, q: p << (q % INT_BITS[t]),
              'shr': lambda p, q: p >> (q % INT_BITS[t]), 'bitand': lambda p, q: p & q, 'bitor': lambda p, q: p | q,
              'bitxor': lambda p, q: p ^ q}[method]
        return Struct('Wrapping', {'0': Int(wrap_int(op(x.v, y.v), t), t)})

    BIN_TRAIT = {'+': 'add', '-': 'sub', '*': 'mul', '/': 'div', '%': 'rem', '<<': 'shl', '>>': 'shr', '&': 'bitand', '|': 'bitor', '^': 'bitxor'}

    def binop(self, op, a, b):
        ta, tb = type(a), type(b)
        if ta is F32 and tb is F32:
            with np.errstate(all='ignore'):
                if op == '*':
                    return a * b
                if op == '+':
                    return a + b
                if op == '-':
                    return a - b
                if op == '/':
                    return a / b
        if isinstance(a, Place):
            a = deref(a)
            ta = type(a)
        if isinstance(b, Place):
            b = deref(b)
            tb = type(b)
        if ta is Int and tb is Int:
            return self.int_binop(op, a, b)
        # untyped float literals
        if ta is ULit or tb is ULit:
            if ta is ULit and tb is ULit:
                if op in ('+', '-', '*', '/', '%'):
                    return ULit(('bin', op, a.tree, b.tree))
                a, b = a.resolve('f64'), b.resolve('f64')
            elif ta is ULit:
                ft = float_type_of(b)
                if ft is None:
                    raise InterpError('float literal combined with %r' % (b,))
                a = a.resolve(ft)
            else:
                ft = float_type_of(a)
                if ft is None:
                    if isinstance(a, Struct):
                        return self.op_trait(a, self.BIN_TRAIT[op], [b])
                    raise InterpError('float literal combined with %r' % (a,))
                b = b.resolve(ft)
            ta, tb = type(a), type(b)
        if ta is Struct:
            if op in self.BIN_TRAIT:
                return self.op_trait(a, self.BIN_TRAIT[op], [b])
            if op == '==':
                return values_equal(a, b)
            if op == '!=':
                return not values_equal(a, b)
        fa, fb = float_type_of(a), float_type_of(b)
        if fa is not None and fb is not None:
            if fa != fb:
                raise InterpError('mismatched float types %s %s %s' % (fa, op, fb))
            with np.errstate(all='ignore'):
                if op == '+':
                    return a + b
                if op == '-':
                    return a - b
                if op == '*':
                    return a * b
                if op == '/':
                    return _fdiv(a, b) if fa == 'f64' else a / b
                if op == '%':
                    return math.fmod(a, b) if fa == 'f64' else F32(math.fmod(float(a), float(b)))
                if op == '<':
                    return a < b
                if op == '>':
                    return a > b
                if op == '<=':
                    return a <= b
                if op == '>=':
                    return a >= b
                if op == '==':
                    return a == b
                if op == '!=':
                    return a != b
            raise InterpError('float op ' + op)
        if isinstance(a, (bool, np.bool_)) and isinstance(b, (bool, np.bool_)):
            a, b = bool(a), bool(b)
            return {'&': a and b, '|': a or b, '^': a != b, '==': a == b, '!=': a != b}[op]
        if op == '==':
            return values_equal(a, b)
        if op == '!=':
            return not values_equal(a, b)
        if isinstance(a, tuple) and isinstance(b, tuple) and op in ('<', '>', '<=', '>='):
            ka, kb = tuple(sort_key(x) for x in a), tuple(sort_key(x) for x in b)
            return {'<': ka < kb, '>': ka > kb, '<=': ka <= kb, '>=': ka >= kb}[op]
        raise InterpError('binary %s on %r and %r' % (op, a, b))

    def int_binop(self, op, a, b):
        t = a.t if a.t is not None else b.t
        if a.t is not None and b.t is not None and a.t != b.t and op not in ('<<', '>>'):
            raise InterpError('mismatched integer types %s %s %s' % (a.t, op, b.t))
        x, y = a.v, b.v
        if op == '+':
            r = x + y
        elif op == '-':
            r = x - y
        elif op == '*':
            r = x * y
        elif op == '/':
            if y == 0:
                raise RustPanic('attempt to divide by zero')
            r = abs(x) // abs(y)
            if (x < 0) != (y < 0):
                r = -r
        elif op == '%':
            if y == 0:
                raise RustPanic('attempt to calculate the remainder with a divisor of zero')
            r = abs(x) % abs(y)
            if x < 0:
                r = -r
        elif op == '<<' or op == '>>':
            t = a.t
            if t is not None:
                bits = INT_BITS[t]
                if not 0 <= y < bits:
                    self.overflows += 1
                    y %= bits
            r = x << y if op == '<<' else x >> y
            if t is None:
                return Int(r)
            return Int(wrap_int(r, t), t)
        elif op == '&':
            r = x & y
        elif op == '|':
            r = x | y
        elif op == '^':
            r = x ^ y
        elif op == '<':
            return x < y
        elif op == '>':
            return x > y
        elif op == '<=':
            return x <= y
        elif op == '>=':
            return x >= y
        elif op == '==':
            return x == y
        elif op == '!=':
            return x != y
        else:
            raise InterpError('int op ' + op)
        if t is None:
            return Int(r)
        w = wrap_int(r, t)
        if w != r:
            self.overflows += 1
        return Int(w, t)

    def e_assign(self, e, env):
        v = self.ev(e[2], env)
        lhs = e[1]
        if lhs[0] == 'tuple':  # destructuring assignment
            for sub, x in zip(lhs[1], v):
                self.place(sub, env).set(x)
            return UNIT
        if lhs[0] == 'path' and lhs[1] == ['_']:
            return UNIT
        p = self.place(lhs, env)
        if isinstance(v, (ULit, Int)):
            try:
                old = p.get()
            except (KeyError, IndexError):
                old = None
            if old is not None and old is not UNINIT:
                v = self.adopt_type(v, old)
        if lhs[0] == 'path':
            cur = p.get() if p.k in p.d else None
            if isinstance(cur, Place) and not isinstance(v, Place):
                # `x = v` where x is a `&mut T` parameter is not valid Rust without `*`; guard against silent rebinding
                pass
        p.set(v)
        return UNIT

    def e_opassign(self, e, env):
        p = self.place(e[2], env)
        rhs = self.ev(e[3], env)
        cur = p.get()
        if isinstance(cur, Place):  # `*x += ..` is spelled with the deref; `x += ..` on a &mut binding auto-derefs for ops
            p = cur
            cur = p.get()
        p.set(self.binop(e[1], cur, rhs))
        return UNIT

    def e_cast(self, e, env):
        v = deref(self.ev(e[1], env))
        ty = e[2]
        if ty[0] == 'tptr' and hasattr(v, 'rs_ptr_cast'):  # a real address (rsinterp/ffi.py CPtr): the cast names its element type
            inner = ty[1]
            name = self.type_name(inner)
            if name in env.generics and isinstance(env.generics[name], tuple) and env.generics[name][0] == 'tpath':
                name = env.generics[name][1][-1]
            return v.rs_ptr_cast(name)
        if ty[0] != 'tpath':
            return v
        name = ty[1][-1]
        if name in env.generics and isinstance(env.generics[name], tuple) and env.generics[name][0] == 'tpath':
            name = env.generics[name][1][-1]
        return self.cast(v, name)

    def cast(self, v, name):
        if name in INT_BITS:
            if isinstance(v, Int):
                return Int(wrap_int(v.v, name), name)
            if isinstance(v, (Arr, Slice)):  # `slice.as_ptr() as usize`: an address -- the identity of the storage
                a, o, _ = seq_view(v)
                return Int(id(a) + o, name)
            if isinstance(v, (bool, np.bool_)):
                return Int(int(v), name)
            if isinstance(v, ULit):
                v = v.resolve('f64')
            if isinstance(v, (float, np.floating)):
                f = float(v)
                if f != f:
                    return Int(0, name)
                bits = INT_BITS[name]
                lo, hi = (0, (1 << bits) - 1) if name[0] == 'u' else (-(1 << (bits - 1)), (1 << (bits - 1)) - 1)
                if f == math.inf:
                    return Int(hi, name)
                if f == -math.inf:
                    return Int(lo, name)
                return Int(min(hi, max(lo, int(f))), name)
            if isinstance(v, Enum):
                return Int(self.discriminant(v), name)
            if isinstance(v, np.ndarray):
                raise InterpError('batched float -> int cast')
            raise InterpError('cast of %r to %s' % (v, name))
        if name == 'f32':
            if isinstance(v, Int):
                return F32(float(v.v)) if abs(v.v) < (1 << 53) else round_to_f32(Fraction(v.v))
            if isinstance(v, ULit):
                return F32(v.resolve('f64'))
            if isinstance(v, np.ndarray):
                return v.astype(np.float32)
            with np.errstate(over='ignore'):
                return F32(v)
        if name == 'f64':
            if isinstance(v, Int):
                return float(v.v)
            if isinstance(v, ULit):
                return v.resolve('f64')
            if isinstance(v, np.ndarray):
                raise InterpError('batched f32 -> f64 cast')
            return float(v)
        if name == 'bool':
            return v
        return v

    def discriminant(self, v):
        nxt = 0
        for var in self.types[v.enum][2]:
            if var[3] is not None:
                nxt = self.ev(var[3], Env()).v
            if var[0] == v.variant:
                return nxt
            nxt += 1
        raise InterpError('discriminant')

    def e_block(self, e, env):
        return self.ev_block(e, env)

    def e_labelled(self, e, env):
        try:
            return self.ev(e[2], env)
        except BreakEx as b:
            if b.label == e[1]:
                return b.value
            raise

    def e_if(self, e, env):
        if truth(self.ev(e[1], env)):
            return self.ev_block(e[2], env)
        if e[3] is not None:
            return self.ev(e[3], env)
        return UNIT

    def e_iflet(self, e, env):
        v = self.evr(e[2], env)
        if isinstance(v, TryInto):
            v = self.resolve_try_into(v, e[1], e[3], env)
        env.push()
        try:
            if self.match(e[1], v, env, True):
                return self.ev_block(e[3], env, new_scope=False)
        finally:
            env.pop()
        if e[4] is not None:
            return self.ev(e[4], env)
        return UNIT

    def resolve_try_into(self, t, pat, then_blk, env):
        """`if let Ok(x) = slice.try_into()`: the target array length is what type inference takes from the first use of
        `x` as an argument of a function whose parameter is a fixed-size array; find that use and compare lengths."""
        if pat[0] == 'ptstruct' and pat[1][-1] == 'Ok' and pat[2] and pat[2][0][0] in ('pident', 'pbind'):
            var = pat[2][0][1]
            n = self.find_array_len_use(then_blk, var, env)
            if n is not None:
                have = seq_view(t.v)[2]
                return ok(t.v) if have == n else err(UNIT)
        raise InterpError('cannot infer the target type of try_into() here')

    def find_array_len_use(self, node, var, env):
        if isinstance(node, tuple):
            if node and node[0] == 'call' and node[1][0] == 'path':
                for i, a in enumerate(node[2]):
                    if a == ('path', [var], None):
                        f = self.resolve_value(node[1][1], env, None)
                        if isinstance(f, FnRef):
                            ty = f.item[3][i][1]
                            while ty[0] == 'tref':
                                ty = ty[2]
                            if ty[0] == 'tarray':
                                return self.ev(ty[2], env).v
            for x in node:
                r = self.find_array_len_use(x, var, env)
                if r is not None:
                    return r
        elif isinstance(node, list):
            for x in node:
                r = self.find_array_len_use(x, var, env)
                if r is not None:
                    return r
        return None

    def e_match(self, e, env):
        v = self.evr(e[1], env)
        if isinstance(v, Place) and not is_agg(v.get()):
            v = v.get()
        for pat, guard, body in e[2]:
            env.push()
            try:
                if self.match(pat, v, env, True):
                    if guard is None or truth(self.ev(guard, env)):
                        return self.ev(body, env)
            finally:
                env.pop()
        raise RustPanic('no match arm matched %r' % (v,))

    def e_while(self, e, env):
        try:
            while truth(self.ev(e[1], env)):
                try:
                    self.ev_block(e[2], env)
                except ContinueEx as c:
                    if c.label is not None:
                        raise
        except BreakEx as b:
            if b.label is not None:
                raise
        return UNIT

    def e_whilelet(self, e, env):
        try:
            while True:
                v = self.evr(e[2], env)
                env.push()
                try:
                    if not self.match(e[1], v, env, True):
                        break
                    try:
                        self.ev_block(e[3], env, new_scope=False)
                    except ContinueEx as c:
                        if c.label is not None:
                            raise
                finally:
                    env.pop()
        except BreakEx as b:
            if b.label is not None:
                raise
        return UNIT

    def e_loop(self, e, env):
        try:
            while True:
                try:
                    self.ev_block(e[1], env)
                except ContinueEx as c:
                    if c.label is not None:
                        raise
        except BreakEx as b:
            if b.label is not None:
                raise
            return b.value

    def e_for(self, e, env):
        it = self.into_iter(self.evr(e[2], env))
        pat, body = e[1], e[3]
        simple = pat[0] == 'pident' and not pat[1][0].isupper()
        try:
            for x in it:
                env.push()
                try:
                    if simple:
                        env.scopes[-1][pat[1]] = x
                    elif not self.match(pat, x, env, True):
                        raise RustPanic('refutable pattern in for')
                    try:
                        self.ev_block(body, env, new_scope=False)
                    except ContinueEx as c:
                        if c.label is not None:
                            raise
                finally:
                    env.pop()
        except BreakEx as b:
            if b.label is not None:
                raise
        return UNIT

    def e_break(self, e, env):
        raise BreakEx(e[1], self.ev(e[2], env) if e[2] is not None else UNIT)

    def e_continue(self, e, env):
        raise ContinueEx(e[1])

    def e_return(self, e, env):
        raise ReturnEx(self.ev(e[1], env) if e[1] is not None else UNIT)

    def e_closure(self, e, env):
        return Closure(e[1], e[2], env)

    def e_try(self, e, env):
        v = deref(self.ev(e[1], env))
        if isinstance(v, Enum):
            if v.variant in ('Ok', 'Some'):
                return v.f['0']
            raise ReturnEx(v)
        raise InterpError('? on %r' % (v,))

    def e_qpath(self, e, env):
        tname = self.type_name(e[1])
        return self.resolve_value([tname] + e[2], env, None)

    def e_call(self, e, env):
        fe = e[1]
        args = [self.ev(a, env) for a in e[2]]
        gargs = None
        if fe[0] == 'path':
            gargs = fe[2]
            segs = fe[1]
            if len(segs) == 2 and segs[0] in ('Self',) and env.self_type:
                segs = [env.self_type, segs[1]]
            if segs[-1] == 'size_of' and gargs and not args:  # std::mem::size_of::<T>() of a primitive
                tn = self.type_name(gargs[0][1]) if isinstance(gargs[0], tuple) and gargs[0][0] == 'gtype' else None
                tn = env.generics.get(tn, tn) if getattr(env, 'generics', None) else tn
                if isinstance(tn, tuple) and tn[0] == 'tpath':  # a type parameter bound by a turbofish (`slot.input::<f32>(0)`)
                    tn = tn[1][-1]
                if not isinstance(tn, str):
                    tn = None
                if tn in INT_BITS:
                    return Int(INT_BITS[tn] // 8, 'usize')
                if tn in FLOAT_TYPES:
                    return Int(4 if tn == 'f32' else 8, 'usize')
                if isinstance(tn, str) and tn in self.types and self.types[tn][0] == 'struct' and getattr(self, 'size_of_struct', None):
                    return Int(self.size_of_struct(tn), 'usize')  # a #[repr(C)] record of the FFI (rsinterp/ffi.py)
                # a type parameter inferred from a declared type the interpreter does not track (Pinned<T>::new): memory
                # is modelled by value (rsinterp/ffi.py RawMem), so a byte count only has to be positive
                return Int(1, 'usize')
            # Type::method(receiver, ..) on builtin types and `f64::sqrt(x)` style calls
            if len(segs) >= 2:
                head = segs[-2]
                if head in FLOAT_TYPES or head in INT_BITS:
                    r = self.prim_assoc(head, segs[-1], args)
                    if r is not NotImplemented:
                        return r
            f = self.evr(('path', segs, gargs), env)
        else:
            f = self.evr(fe, env)
        if isinstance(f, Place):
            f = f.get()
        return self.call_value(f, args, gargs, env)

    def prim_assoc(self, ty, name, args):
        if name == 'from' or name == 'try_from':
            v = deref(args[0])
            if ty in INT_BITS:
                iv = v.v if isinstance(v, Int) else int(v)
                if name == 'try_from':
                    return ok(Int(iv, ty)) if in_range(iv, ty) else err(UNIT)
                return Int(iv, ty)
            return self.cast(v, ty)
        if name in ('max_value', 'MAX') and not args:
            return self.prim_const(ty, 'MAX')
        if name in ('min_value', 'MIN') and not args:
            return self.prim_const(ty, 'MIN')
        if name == 'from_bits':
            v = deref(args[0])
            if ty == 'f32':
                return np.uint32(v.v).view(np.float32)
            return _struct.unpack('<d', _struct.pack('<Q', v.v))[0]
        if name in ('from_le_bytes', 'from_be_bytes', 'from_ne_bytes'):
            bs = bytes(x.v for x in seq_list(args[0]))
            return Int(int.from_bytes(bs, 'big' if name == 'from_be_bytes' else 'little', signed=ty[0] == 'i'), ty)
        # f64::sqrt(x), u32::min(a, b), ... : the method form
        if args:
            recv = deref(args[0])
            if isinstance(recv, ULit):
                recv = recv.resolve(ty)
            if isinstance(recv, Int) and recv.t is None and ty in INT_BITS:
                recv = Int(recv.v, ty)
            return self.builtin_method(recv, name, list(args[1:]), None, None)
        return NotImplemented

    def prim_const(self, ty, name):
        if ty in INT_BITS:
            bits = INT_BITS[ty]
            if name == 'MAX':
                return Int((1 << bits) - 1 if ty[0] == 'u' else (1 << (bits - 1)) - 1, ty)
            if name == 'MIN':
                return Int(0 if ty[0] == 'u' else -(1 << (bits - 1)), ty)
            if name == 'BITS':
                return Int(bits, 'u32')
        consts = {'PI': math.pi, 'E': math.e, 'SQRT_2': math.sqrt(2.0), 'FRAC_1_SQRT_2': 0.70710678118654752440, 'FRAC_PI_2': math.pi / 2,
                  'FRAC_PI_4': math.pi / 4, 'FRAC_PI_3': math.pi / 3, 'FRAC_PI_6': math.pi / 6, 'FRAC_PI_8': math.pi / 8, 'TAU': 2 * math.pi,
                  'LN_2': math.log(2.0), 'LN_10': math.log(10.0), 'LOG2_E': 1.4426950408889634, 'LOG10_E': 0.4342944819032518,
                  'FRAC_1_PI': 0.3183098861837907, 'FRAC_2_PI': 0.6366197723675814, 'FRAC_2_SQRT_PI': 1.1283791670955126,
                  'LOG2_10': 3.321928094887362, 'LOG10_2': 0.3010299956639812}
        # std's f32 constants are the f32 literals nearest to the real values = the rounded f64 value for each of these
        F32_TEXT = {'PI': '3.14159265358979323846264338327950288', 'FRAC_1_SQRT_2': '0.707106781186547524400844362104849039',
                    'SQRT_2': '1.41421356237309504880168872420969808', 'E': '2.71828182845904523536028747135266250',
                    'FRAC_PI_2': '1.57079632679489661923132169163975144', 'FRAC_PI_4': '0.785398163397448309615660845819875721',
                    'TAU': '6.28318530717958647692528676655900577', 'LN_2': '0.693147180559945309417232121458176568',
                    'LN_10': '2.30258509299404568401799145468436421', 'FRAC_PI_3': '1.04719755119659774615421446109316763',
                    'FRAC_PI_6': '0.52359877559829887307710723054658381', 'FRAC_PI_8': '0.39269908169872415480783042290993786'}
        if name in consts:
            if ty == 'f64':
                return consts[name]
            if name in F32_TEXT:
                return f32_from_text(F32_TEXT[name])
            return F32(consts[name])
        if ty in FLOAT_TYPES:
            fi = np.finfo(np.float32 if ty == 'f32' else np.float64)
            tab = {'MAX': fi.max, 'MIN': fi.min, 'EPSILON': fi.eps, 'MIN_POSITIVE': fi.tiny, 'INFINITY': np.inf, 'NEG_INFINITY': -np.inf, 'NAN': np.nan}
            if name in tab:
                return F32(tab[name]) if ty == 'f32' else float(tab[name])
        raise InterpError('unknown constant %s::%s' % (ty, name))

    # ---------------------------------------------------------------- built-in paths
    def builtin_fn(self, segs):
        name = segs[-1]
        head = segs[-2] if len(segs) >= 2 else None
        from . import stdext
        ext = stdext.path_builtin(self, segs)
        if ext is not None:
            return ext
        if head == 'consts' and len(segs) >= 3 and segs[-3] in FLOAT_TYPES:
            return self.prim_const(segs[-3], name)
        if head in FLOAT_TYPES or head in INT_BITS:
            if name in ('MAX', 'MIN', 'BITS', 'EPSILON', 'INFINITY', 'NEG_INFINITY', 'NAN', 'MIN_POSITIVE'):
                return self.prim_const(head, name)
            return Builtin(lambda *a, _h=head, _n=name: self.prim_assoc(_h, _n, list(a)), head + '::' + name)
        if name == 'Some':
            return Builtin(some, 'Some')
        if name == 'None':
            return NONE
        if name == 'Ok':
            return Builtin(ok, 'Ok')
        if name == 'Err':
            return Builtin(err, 'Err')
        if name == 'Wrapping':
            return Builtin(lambda v: Struct('Wrapping', {'0': deref(v)}), 'Wrapping')
        if head in ('Box', 'Rc', 'Arc', 'Cell', 'RefCell') and name == 'new':
            return Builtin(lambda v: v, 'Box::new')
        if head == 'Vec':
            if name == 'new':
                return Builtin(lambda: Arr([], True), 'Vec::new')
            if name == 'with_capacity':
                return Builtin(lambda n: Arr([], True), 'Vec::with_capacity')
            if name == 'from':
                return Builtin(lambda v: Arr(list(seq_list(deepclone(v))), True), 'Vec::from')
        if head == 'Default' and name == 'default' or (head is None and name == 'default'):
            return Builtin(lambda: UNINIT, 'Default::default')
        if head in ('cmp', 'std') or head is None:
            if name == 'min':
                return Builtin(lambda a, b: self.builtin_method(deref(a), 'min', [b], None, None), 'min')
            if name == 'max':
                return Builtin(lambda a, b: self.builtin_method(deref(a), 'max', [b], None, None), 'max')
        if head == 'mem':
            if name == 'swap':
                def swap(a, b):
                    pa = a if isinstance(a, Place) else ObjPlace(a)
                    pb = b if isinstance(b, Place) else ObjPlace(b)
                    va, vb = copyval(pa.get()), copyval(pb.get())
                    pa.set(vb)
                    pb.set(va)
                return Builtin(swap, 'mem::swap')
            if name == 'replace':
                def replace(a, v):
                    pa = a if isinstance(a, Place) else ObjPlace(a)
                    old = copyval(pa.get())
                    pa.set(v)
                    return old
                return Builtin(replace, 'mem::replace')
            if name == 'take':
                raise InterpError('mem::take needs a type')
        if head == 'iter' and name == 'repeat_n':
            return Builtin(lambda v, n: RIter(lst=[deref(v)] * deref(n).v), 'iter::repeat_n')
        if head == 'iter' and name == 'repeat':
            def rep(v):
                def g():
                    while True:
                        yield v
                return RIter(g())
            return Builtin(rep, 'iter::repeat')
        if name == 'drop':
            return Builtin(lambda v: UNIT, 'drop')
        return None

    # ---------------------------------------------------------------- macros
    def e_macro(self, e, env):
        name, toks = e[1], e[2]
        if name in ('assert', 'debug_assert'):
            parts = self.macro_args(e)
            if not truth(self.ev(parts[0], env)):
                raise RustPanic('assertion failed: ' + ' '.join(t.s for t in P.split_commas(toks)[0]))
            return UNIT
        if name in ('assert_eq', 'debug_assert_eq', 'assert_ne', 'debug_assert_ne'):
            parts = self.macro_args(e)
            a, b = self.ev(parts[0], env), self.ev(parts[1], env)
            eq = values_equal(a, b)
            if eq != ('eq' in name):
                raise RustPanic('assertion `left %s right` failed: %r vs %r' % ('==' if 'eq' in name else '!=', a, b))
            return UNIT
        if name == 'vec':
            key = id(e)
            node = self.node_cache.get(key)
            if node is None:
                p = P.Parser(list(toks) + [P.Tok('eof', '', 0)])
                if p.cur.k == 'eof':
                    node = ('array', [])
                else:
                    first = p.expr()
                    if p.eat(';'):
                        node = ('repeat', first, p.expr())
                    else:
                        elems = [first]
                        while p.eat(','):
                            if p.cur.k == 'eof':
                                break
                            elems.append(p.expr())
                        node = ('array', elems)
                self.node_cache[key] = node
            v = self.ev(node, env)
            v.vec = True
            return v
        if name in ('panic', 'unreachable', 'unimplemented', 'todo'):
            raise RustPanic(name + '!: ' + ' '.join(t.s for t in toks))
        if name in ('println', 'print', 'eprintln', 'eprint', 'debug', 'info', 'warn', 'error', 'trace', 'log', 'dbg'):
            return UNIT
        if name == 'matches':
            key = id(e)
            node = self.node_cache.get(key)
            if node is None:
                parts = P.split_commas(toks)
                ex = P.parse_tokens_as_expr(parts[0])
                pp = P.Parser(parts[1] + [P.Tok('eof', '', 0)])
                node = (ex, pp.pattern())
                self.node_cache[key] = node
            v = self.evr(node[0], env)
            env.push()
            try:
                return self.match(node[1], v, env, True)
            finally:
                env.pop()
        if name in ('format', 'concat', 'stringify', 'line', 'file', 'column'):
            return ''
        if name == 'cfg':  # no configuration flag is set here (not fuzzing, no debug assertions, no optional feature)
            return bool(toks) and toks[0].s == 'not'
        if name in self.macros:
            key = id(e)
            node = self.node_cache.get(key)
            if node is None:
                node = P.parse_tokens_as_expr(self.expand_macro(name, toks))
                self.node_cache[key] = node
            return self.ev(node, env)
        raise InterpError('macro %s! is not supported' % name)

    def macro_args(self, e):
        key = id(e)
        node = self.node_cache.get(key)
        if node is None:
            node = [P.parse_tokens_as_expr(p) for p in P.split_commas(e[2]) if p]
            self.node_cache[key] = node
        return node

    # ---------------------------------------------------------------- method calls
    def e_mcall(self, e, env):
        _, recv_e, name, gargs, arg_es = e
        recv = self.evr(recv_e, env)
        base = deref(recv)
        from . import stdext
        while stdext.is_std(base):
            done, val = stdext.method(self, base, name, [self.ev(a, env) for a in arg_es] if not isinstance(base, stdext.Cell) or base.kind != 'Arc' or name == 'clone' else [], env)
            if done:
                return val
            base = recv = val  # Arc<T>: the method is T's
        if hasattr(base, 'rs_method'):  # raw pointers, C strings (rsinterp/ffi.py)
            return base.rs_method(self, name, [self.ev(a, env) for a in arg_es])
        # user-defined methods
        if isinstance(base, (Struct, Enum)):
            tname = base.name if isinstance(base, Struct) else base.enum
            native = self.native_methods.get((tname, name))
            if native is not None:  # a method implemented by the harness in Python (tests: FFI bridges, type-tagged views)
                return native(base, [self.ev(a, env) for a in arg_es])
            m = self.impls.get(tname, {}).get(name)
            if m is None:
                m = self.trait_default(tname, name)
            if m is not None:
                args = [self.ev(a, env) for a in arg_es]
                self_val = copyval(base) if m[4] == 'value' else ObjPlace(base)
                return self.call_fn(m, args, gargs, tname, env, self_val=self_val)
        args = [self.ev(a, env) for a in arg_es]
        return self.builtin_method(base, name, args, recv, env, gargs)

    def trait_default(self, tname, name):
        """the default body of method `name` in a trait `tname` implements (or in one of that trait's supertraits)"""
        seen, todo = set(), list(self.trait_impls.get(tname, []))
        while todo:
            tr = todo.pop(0)
            if tr in seen:
                continue
            seen.add(tr)
            m = self.impls.get('<trait ' + tr + '>', {}).get(name)
            if m is not None:
                return m
            todo.extend(self.supertraits.get(tr, []))
        return None

    def into_iter(self, v):
        if isinstance(v, ObjPlace) and v.mut and isinstance(v.o, Arr):      # for x in &mut array
            return RIter(lst=[ElemPlace(v.o.a, i) for i in range(len(v.o.a))])
        v = deref(v)
        if isinstance(v, Slice) and v.mut:                                   # for x in a `&mut [T]`
            return RIter(lst=[ElemPlace(v.a, i) for i in range(v.o, v.o + v.n)])
        if isinstance(v, RIter):
            return v
        if isinstance(v, Range):
            return RIter(lst=self.range_list(v))
        if isinstance(v, (Arr, Slice)):
            a, o, n = seq_view(v)
            return RIter(lst=a[o:o + n])
        if isinstance(v, Enum) and v.enum == 'Option':
            return RIter(lst=[v.f['0']] if v.variant == 'Some' else [])
        if isinstance(v, Struct) and 'next' in self.impls.get(v.name, {}):
            nxt = self.impls[v.name]['next']

            def g():
                while True:
                    r = self.call_fn(nxt, [], None, v.name, None, self_val=ObjPlace(v))
                    if r.variant == 'None':
                        return
                    yield r.f['0']
            return RIter(g())
        raise InterpError('not iterable: %r' % (v,))

    def range_list(self, r):
        if r.hi is None:
            def g():
                i = r.lo.v
                while True:
                    yield Int(i, r.lo.t)
                    i += 1
            return IterList(g())
        t = r.lo.t if r.lo.t is not None else r.hi.t
        hi = r.hi.v + (1 if r.incl else 0)
        return [Int(i, t) for i in range(r.lo.v, hi)]

    def iter_mut_list(self, v):
        a, o, n = seq_view(v)
        return [a[i] if isinstance(a[i], (Arr, Struct)) and False else ElemPlace(a, i) for i in range(o, o + n)]

    def iter_list(self, v):
        a, o, n = seq_view(v)
        return a[o:o + n]

    def builtin_method(self, base, name, args, recv, env, gargs=None):
        tb = type(base)
        if tb is Int:
            return self.int_method(base, name, args)
        if tb is F32 or tb is float or tb is np.ndarray:
            return self.float_method(base, name, args)
        if tb is ULit:
            ft = None
            for a in args:
                ft = ft or float_type_of(deref(a))
            return self.float_method(base.resolve(ft or 'f64'), name, args)
        if tb is Arr or tb is Slice:
            return self.seq_method(base, name, args, recv, env, gargs)
        if tb is RIter:
            return self.iter_method(base, name, args, env, gargs)
        if tb is Range:
            if name in ('contains',):
                x = deref(args[0])
                xv = x.v if isinstance(x, Int) else x
                lo = base.lo.v if isinstance(base.lo, Int) else base.lo
                hi = base.hi.v if isinstance(base.hi, Int) else base.hi
                return (lo is None or xv >= lo) and (hi is None or (xv <= hi if base.incl else xv < hi))
            if name == 'len':
                return Int(len(self.range_list(base)), 'usize')
            if name in ('start', 'end'):
                return base.lo if name == 'start' else base.hi
            if name in ('clone', 'into_iter', 'iter'):
                return base if name == 'clone' else self.into_iter(base)
            return self.iter_method(self.into_iter(base), name, args, env, gargs)
        if tb is Enum:
            return self.enum_method(base, name, args, env)
        if tb is tuple:
            if name == 'clone':
                return copyval(base)
        if tb is bool or tb is np.bool_:
            if name == 'then':
                return some(self.call_value(args[0], [])) if base else NONE
            if name == 'then_some':
                return some(args[0]) if base else NONE
            if name in ('clone', 'into'):
                return bool(base)
        if tb is Struct:
            if 'next' in self.impls.get(base.name, {}):
                return self.iter_method(self.into_iter(base), name, args, env, gargs)
            if base.name == 'Wrapping' and name in ('clone',):
                return copyval(base)
            if name in ('clone',):
                return deepclone(base)
            if name in ('as_ref', 'as_mut', 'borrow', 'borrow_mut', 'deref', 'deref_mut', 'into'):
                return base
        if tb is TryInto:
            if name in ('unwrap', 'expect'):
                return base.v
            if name == 'ok':
                return some(base.v)
        if tb is str:
            if name in ('len',):
                return Int(len(base), 'usize')
            return base
        if tb is Closure or tb is FnRef:
            if name in ('clone', 'borrow'):
                return base
        if tb is Uninit:
            raise InterpError('use of a Default::default() placeholder (method %s)' % name)
        raise InterpError('no method %s on %r' % (name, base))

    def int_method(self, x, name, args):
        t = x.t
        a = [deref(v) for v in args]

        def other(i=0):
            o = a[i]
            if isinstance(o, Int):
                return o.v
            raise InterpError('int method %s with %r' % (name, o))

        def ty():
            if t is not None:
                return t
            for o in a:
                if isinstance(o, Int) and o.t is not None:
                    return o.t
            return 'i32'
        if name.startswith('wrapping_'):
            op = name[9:]
            T = ty()
            bits = INT_BITS[T]
            if op == 'neg':
                return Int(wrap_int(-x.v, T), T)
            if op == 'abs':
                return Int(wrap_int(abs(x.v), T), T)
            if op in ('shl', 'shr'):
                s = other() % bits
                return Int(wrap_int(x.v << s if op == 'shl' else x.v >> s, T), T)
            r = {'add': x.v + other(), 'sub': x.v - other(), 'mul': x.v * other()}.get(op)
            if r is None:
                if op == 'div':
                    return self.int_binop('/', Int(x.v, T), Int(other(), T))
                if op == 'rem':
                    return self.int_binop('%', Int(x.v, T), Int(other(), T))
                if op == 'pow':
                    r = x.v ** other()
                else:
                    raise InterpError('int method ' + name)
            return Int(wrap_int(r, T), T)
        if name.startswith('checked_') or name.startswith('saturating_') or name.startswith('overflowing_'):
            kind, op = name.split('_', 1)
            T = ty()
            if op in ('add', 'sub', 'mul', 'pow'):
                r = {'add': lambda: x.v + other(), 'sub': lambda: x.v - other(), 'mul': lambda: x.v * other(), 'pow': lambda: x.v ** other()}[op]()
            elif op in ('div', 'rem'):
                if other() == 0:
                    return NONE
                r = self.int_binop('/' if op == 'div' else '%', Int(x.v), Int(other())).v
            elif op in ('shl', 'shr'):
                if not 0 <= other() < INT_BITS[T]:
                    return NONE
                r = wrap_int(x.v << other(), T) if op == 'shl' else x.v >> other()
            elif op == 'neg':
                r = -x.v
            elif op == 'abs':
                r = abs(x.v)
            else:
                raise InterpError('int method ' + name)
            fits = in_range(r, T)
            if kind == 'checked':
                return some(Int(r, T)) if fits else NONE
            if kind == 'overflowing':
                return (Int(wrap_int(r, T), T), not fits)
            if fits:
                return Int(r, T)
            bits = INT_BITS[T]
            lo, hi = (0, (1 << bits) - 1) if T[0] == 'u' else (-(1 << (bits - 1)), (1 << (bits - 1)) - 1)
            return Int(hi if r > hi else lo, T)
        if name in ('min', 'max'):
            o = a[0]
            T = ty()
            r = min(x.v, o.v) if name == 'min' else max(x.v, o.v)
            return Int(r, t if t is not None else o.t)
        if name == 'clamp':
            return Int(min(max(x.v, a[0].v), a[1].v), ty())
        if name == 'pow':
            r = x.v ** other()
            if t is None:
                return Int(r)
            w = wrap_int(r, t)
            if w != r:
                self.overflows += 1
            return Int(w, t)
        if name == 'abs':
            return Int(abs(x.v), t)
        if name == 'unsigned_abs':
            return Int(abs(x.v), 'u' + ty()[1:])
        if name == 'abs_diff':
            return Int(abs(x.v - other()), 'u' + ty()[1:])
        if name == 'signum':
            return Int((x.v > 0) - (x.v < 0), t)
        if name == 'is_power_of_two':
            return x.v > 0 and (x.v & (x.v - 1)) == 0
        if name == 'next_power_of_two':
            return Int(1 if x.v <= 1 else 1 << (x.v - 1).bit_length(), t)
        if name == 'leading_zeros':
            bits = INT_BITS[ty()]
            return Int(bits - (x.v & ((1 << bits) - 1)).bit_length(), 'u32')
        if name == 'trailing_zeros':
            bits = INT_BITS[ty()]
            v = x.v & ((1 << bits) - 1)
            return Int(bits if v == 0 else (v & -v).bit_length() - 1, 'u32')
        if name == 'get' and not a:  # NonZero<T>::get
            return x
        if name in ('leading_ones', 'trailing_ones'):
            bits = INT_BITS[ty()]
            v = ~x.v & ((1 << bits) - 1)
            if name == 'leading_ones':
                return Int(bits - v.bit_length(), 'u32')
            return Int(bits if v == 0 else (v & -v).bit_length() - 1, 'u32')
        if name in ('count_ones', 'count_zeros'):
            bits = INT_BITS[ty()]
            ones = bin(x.v & ((1 << bits) - 1)).count('1')
            return Int(ones if name == 'count_ones' else bits - ones, 'u32')
        if name == 'ilog2':
            return Int(x.v.bit_length() - 1, 'u32')
        if name == 'reverse_bits':
            bits = INT_BITS[ty()]
            v = x.v & ((1 << bits) - 1)
            return Int(wrap_int(int(format(v, '0%db' % bits)[::-1], 2), ty()), ty())
        if name == 'swap_bytes':
            bits = INT_BITS[ty()]
            v = (x.v & ((1 << bits) - 1)).to_bytes(bits // 8, 'little')
            return Int(wrap_int(int.from_bytes(v, 'big'), ty()), ty())
        if name in ('rotate_left', 'rotate_right'):
            bits = INT_BITS[ty()]
            v = x.v & ((1 << bits) - 1)
            s = other() % bits
            if name == 'rotate_right':
                s = (bits - s) % bits
            return Int(wrap_int(((v << s) | (v >> (bits - s))) & ((1 << bits) - 1), ty()), ty())
        if name == 'div_ceil':
            return Int(-((-x.v) // other()), t)
        if name == 'div_euclid':
            return Int(x.v // other() if other() > 0 else -(x.v // -other()), t)
        if name == 'rem_euclid':
            return Int(x.v % abs(other()), t)
        if name in ('clone', 'into', 'to_owned', 'borrow', 'get', 'as_ref'):
            return x
        if name in ('try_into',):
            return TryInto(x)
        if name in ('is_positive', 'is_negative'):
            return x.v > 0 if name == 'is_positive' else x.v < 0
        if name in ('to_le_bytes', 'to_be_bytes', 'to_ne_bytes'):
            bits = INT_BITS[ty()]
            bs = (x.v & ((1 << bits) - 1)).to_bytes(bits // 8, 'big' if name == 'to_be_bytes' else 'little')
            return Arr([Int(b, 'u8') for b in bs])
        if name in ('eq', 'ne', 'lt', 'le', 'gt', 'ge'):
            o = other()
            return {'eq': x.v == o, 'ne': x.v != o, 'lt': x.v < o, 'le': x.v <= o, 'gt': x.v > o, 'ge': x.v >= o}[name]
        if name == 'cmp':
            o = other()
            return Enum('Ordering', 'Less' if x.v < o else 'Greater' if x.v > o else 'Equal')
        if name == 'is_ascii_digit':
            return 48 <= x.v <= 57
        raise InterpError('no int method %s' % name)

    def float_method(self, x, name, args):
        is32 = not isinstance(x, float)
        a = []
        for v in args:
            v = deref(v)
            if isinstance(v, ULit):
                v = v.resolve('f32' if is32 else 'f64')
            a.append(v)

        def lib(fn32, fn64):
            if is32:
                if isinstance(x, np.ndarray):
                    return np.array([fn32(ctypes.c_float(float(v))) for v in x], dtype=np.float32)
                return F32(fn32(ctypes.c_float(float(x))))
            return fn64(x)
        with np.errstate(all='ignore'):
            if name == 'abs':
                return np.abs(x) if is32 else abs(x)
            if name == 'sqrt':
                if is32:
                    return np.sqrt(x)
                return math.sqrt(x) if x >= 0 else math.nan
            if name == 'sin':
                return lib(_libm.sinf, math.sin)
            if name == 'cos':
                return lib(_libm.cosf, math.cos)
            if name == 'tan':
                return lib(_libm.tanf, math.tan)
            if name == 'exp':
                return lib(_libm.expf, math.exp)
            if name == 'exp2':
                return lib(_libm.exp2f, _libm.exp2)
            if name == 'ln':
                return lib(_libm.logf, lambda v: math.log(v) if v > 0 else (-math.inf if v == 0 else math.nan))
            if name == 'log2':
                return lib(_libm.log2f, _libm.log2)
            if name == 'log10':
                return lib(_libm.log10f, lambda v: math.log10(v) if v > 0 else (-math.inf if v == 0 else math.nan))
            if name == 'atan':
                return lib(_libm.atanf, math.atan)
            if name == 'asin':
                return lib(_libm.asinf, math.asin)
            if name == 'acos':
                return lib(_libm.acosf, math.acos)
            if name == 'sinh':
                return lib(_libm.sinhf, math.sinh)
            if name == 'cosh':
                return lib(_libm.coshf, math.cosh)
            if name == 'tanh':
                return lib(_libm.tanhf, math.tanh)
            if name == 'powf':
                if is32:
                    return F32(_libm.powf(ctypes.c_float(float(x)), ctypes.c_float(float(a[0]))))
                try:
                    return math.pow(x, a[0])
                except (OverflowError, ValueError):
                    return math.inf if abs(x) > 1 else math.nan
            if name == 'powi':
                # compiler-rt __powisf2 / __powidf2: square-and-multiply, reciprocal for negative exponents
                n = a[0].v
                b = x
                recip = n < 0
                n = abs(n)
                r = F32(1.0) if is32 else 1.0
                while True:
                    if n & 1:
                        r = r * b
                    n >>= 1
                    if n == 0:
                        break
                    b = b * b
                return (F32(1.0) / r if is32 else _fdiv(1.0, r)) if recip else r
            if name == 'atan2':
                return F32(_libm.atan2f(ctypes.c_float(float(x)), ctypes.c_float(float(a[0])))) if is32 else math.atan2(x, a[0])
            if name == 'hypot':
                return F32(_libm.hypotf(ctypes.c_float(float(x)), ctypes.c_float(float(a[0])))) if is32 else math.hypot(x, a[0])
            if name == 'floor':
                return np.floor(x) if is32 else float(math.floor(x)) if math.isfinite(x) else x
            if name == 'ceil':
                return np.ceil(x) if is32 else float(math.ceil(x)) if math.isfinite(x) else x
            if name == 'trunc':
                return np.trunc(x) if is32 else float(math.trunc(x)) if math.isfinite(x) else x
            if name == 'round':  # half away from zero
                if is32:
                    return np.copysign(np.floor(np.abs(x) + F32(0.5)), x) if np.all(np.abs(x) < 8388608) else x
                return math.copysign(math.floor(abs(x) + 0.5), x) if abs(x) < 4503599627370496.0 else x
            if name == 'fract':
                return x - (np.trunc(x) if is32 else float(math.trunc(x)))
            if name in ('min', 'max'):
                o = a[0]
                if is32:
                    return np.fmin(x, o) if name == 'min' else np.fmax(x, o)
                if x != x:
                    return o
                if o != o:
                    return x
                return min(x, o) if name == 'min' else max(x, o)
            if name == 'clamp':
                lo, hi = a
                if is32:
                    return np.minimum(np.maximum(x, lo), hi)
                return min(max(x, lo), hi)
            if name == 'mul_add':
                r = Fraction(float(x)) * Fraction(float(a[0])) + Fraction(float(a[1]))
                return round_to_f32(r) if is32 else float(r)
            if name == 'recip':
                return F32(1.0) / x if is32 else _fdiv(1.0, x)
            if name == 'signum':
                return np.copysign(F32(1.0), x) if is32 else math.copysign(1.0, x)
            if name == 'copysign':
                return np.copysign(x, a[0]) if is32 else math.copysign(x, a[0])
            if name == 'is_nan':
                return np.isnan(x) if is32 else x != x
            if name == 'is_infinite':
                return np.isinf(x) if is32 else math.isinf(x)
            if name == 'is_finite':
                return np.isfinite(x) if is32 else math.isfinite(x)
            if name == 'is_sign_positive':
                return not np.signbit(x) if is32 else math.copysign(1.0, x) > 0
            if name == 'is_sign_negative':
                return bool(np.signbit(x)) if is32 else math.copysign(1.0, x) < 0
            if name == 'to_bits':
                if is32:
                    return Int(int(F32(x).view(np.uint32)), 'u32')
                return Int(_struct.unpack('<Q', _struct.pack('<d', x))[0], 'u64')
            if name == 'to_degrees':
                return x * (F32(180.0) / f32_from_text('3.14159265358979323846264338327950288')) if is32 else x * (180.0 / math.pi)
            if name == 'to_radians':
                return x * (f32_from_text('3.14159265358979323846264338327950288') / F32(180.0)) if is32 else x * (math.pi / 180.0)
            if name in ('clone', 'into', 'to_owned', 'borrow'):
                return x
            if name in ('partial_cmp', 'total_cmp'):
                o = a[0]
                ordv = Enum('Ordering', 'Less' if x < o else 'Greater' if x > o else 'Equal')
                return some(ordv) if name == 'partial_cmp' else ordv
            if name in ('lt', 'le', 'gt', 'ge', 'eq', 'ne'):
                o = a[0]
                return {'lt': x < o, 'le': x <= o, 'gt': x > o, 'ge': x >= o, 'eq': x == o, 'ne': x != o}[name]
        raise InterpError('no float method %s' % name)

    def seq_method(self, v, name, args, recv, env, gargs):
        a, o, n = seq_view(v)
        if name == 'len':
            return Int(n, 'usize')
        if name == 'is_empty':
            return n == 0
        if name in ('iter', 'into_iter'):
            return RIter(lst=a[o:o + n])
        if name == 'iter_mut':
            return RIter(lst=[ElemPlace(a, i) for i in range(o, o + n)])
        if name in ('as_ref', 'as_mut', 'as_slice', 'as_mut_slice', 'borrow', 'borrow_mut', 'into_boxed_slice', 'into', 'as_ptr', 'as_mut_ptr', 'deref',
                    'deref_mut', 'into_vec'):
            if name in ('into_boxed_slice', 'into_vec') and isinstance(v, Arr):
                v.vec = True
                return v
            return ObjPlace(v) if isinstance(v, Arr) and name not in ('into',) else v
        if name in ('to_vec', 'clone', 'to_owned'):
            return Arr([copyval(x) for x in a[o:o + n]], name != 'clone' or (isinstance(v, Arr) and v.vec) or isinstance(v, Slice))
        if name == 'try_into':
            return TryInto(recv if isinstance(recv, Place) else v)  # a reference stays a reference
        if name == 'fill':
            val = args[0]
            if n and isinstance(val, (ULit, Int)):
                val = self.adopt_type(val, a[o])
            for i in range(o, o + n):
                a[i] = copyval(val) if is_agg(val) else val
            return UNIT
        if name in ('copy_from_slice', 'clone_from_slice'):
            sa, so, sn = seq_view(args[0])
            if sn != n:
                raise RustPanic('source slice length (%d) does not match destination slice length (%d)' % (sn, n))
            a[o:o + n] = [copyval(x) for x in sa[so:so + sn]]
            return UNIT
        if name == 'copy_within':
            r = deref(args[0])
            lo = r.lo.v if r.lo is not None else 0
            hi = (r.hi.v + (1 if r.incl else 0)) if r.hi is not None else n
            dst = deref(args[1]).v
            a[o + dst:o + dst + hi - lo] = a[o + lo:o + hi]
            return UNIT
        if name == 'swap':
            i, j = deref(args[0]).v, deref(args[1]).v
            if not (0 <= i < n and 0 <= j < n):
                raise RustPanic('swap index out of bounds')
            a[o + i], a[o + j] = a[o + j], a[o + i]
            return UNIT
        if name == 'reverse':
            a[o:o + n] = a[o:o + n][::-1]
            return UNIT
        if name in ('split_at', 'split_at_mut'):
            m = deref(args[0]).v
            if m > n:
                raise RustPanic('mid > len')
            mt = name.endswith('_mut')
            return (Slice(a, o, m, mt), Slice(a, o + m, n - m, mt))
        if name in ('split_first', 'split_first_mut', 'split_last', 'split_last_mut'):
            if n == 0:
                return NONE
            first = name.startswith('split_first')
            el = ElemPlace(a, o if first else o + n - 1) if name.endswith('_mut') else a[o if first else o + n - 1]
            return some((el, Slice(a, o + 1, n - 1) if first else Slice(a, o, n - 1)))
        if name in ('chunks', 'chunks_mut', 'chunks_exact', 'chunks_exact_mut'):
            c = deref(args[0]).v
            if c == 0:
                raise RustPanic('chunk size must be non-zero')
            exact = 'exact' in name
            out = []
            i = 0
            while i < n:
                ln = min(c, n - i)
                if exact and ln < c:
                    break
                out.append(Slice(a, o + i, ln, name.endswith('_mut')))
                i += c
            return RIter(lst=out, rem=Slice(a, o + i, n - i if exact and i < n else 0, name.endswith('_mut')))
        if name == 'windows':
            c = deref(args[0]).v
            return RIter(lst=[Slice(a, o + i, c) for i in range(0, n - c + 1)])
        if name in ('first', 'last', 'first_mut', 'last_mut'):
            if n == 0:
                return NONE
            i = o if name.startswith('first') else o + n - 1
            return some(ElemPlace(a, i) if name.endswith('mut') else a[i])
        if name in ('get', 'get_mut'):
            idx = deref(args[0])
            if isinstance(idx, Int):
                if 0 <= idx.v < n:
                    return some(ElemPlace(a, o + idx.v) if name == 'get_mut' else a[o + idx.v])
                return NONE
            try:
                return some(self.index_value(v, idx))
            except RustPanic:
                return NONE
        if name == 'contains':
            return any(values_equal(x, args[0]) for x in a[o:o + n])
        if name in ('sort', 'sort_unstable'):
            a[o:o + n] = sorted(a[o:o + n], key=sort_key)
            return UNIT
        if name in ('sort_by_key', 'sort_unstable_by_key'):
            a[o:o + n] = sorted(a[o:o + n], key=lambda x: sort_key(self.call_value(args[0], [x])))
            return UNIT
        if name == 'concat':
            out = []
            for x in a[o:o + n]:
                out.extend(seq_list(x))
            return Arr(out, True)
        if name == 'rotate_left':
            k = deref(args[0]).v
            a[o:o + n] = a[o + k:o + n] + a[o:o + k]
            return UNIT
        if name == 'rotate_right':
            k = deref(args[0]).v
            a[o:o + n] = a[o + n - k:o + n] + a[o:o + n - k]
            return UNIT
        # Vec
        if isinstance(v, Arr):
            if name in ('push', 'push_back'):
                v.a.append(args[0])
                return UNIT
            if name in ('pop', 'pop_back'):
                return some(v.a.pop()) if v.a else NONE
            if name == 'pop_front':
                return some(v.a.pop(0)) if v.a else NONE
            if name == 'push_front':
                v.a.insert(0, args[0])
                return UNIT
            if name in ('front', 'front_mut'):
                return some(v.a[0]) if v.a else NONE
            if name in ('back', 'back_mut'):
                return some(v.a[-1]) if v.a else NONE
            if name == 'retain':
                v.a[:] = [x for x in v.a if truth(self.call_value(args[0], [x]))]
                return UNIT
            if name == 'clear':
                v.a.clear()
                return UNIT
            if name == 'truncate':
                del v.a[deref(args[0]).v:]
                return UNIT
            if name == 'resize':
                m = deref(args[0]).v
                val = args[1]
                if m < len(v.a):
                    del v.a[m:]
                else:
                    v.a.extend(copyval(val) if is_agg(val) else val for _ in range(m - len(v.a)))
                return UNIT
            if name in ('extend', 'extend_from_slice', 'append'):
                src = self.into_iter(args[0])
                v.a.extend(copyval(deref(x)) if is_agg(deref(x)) else deref(x) for x in src)
                return UNIT
            if name == 'insert':
                v.a.insert(deref(args[0]).v, args[1])
                return UNIT
            if name == 'remove':
                return v.a.pop(deref(args[0]).v)
            if name in ('reserve', 'reserve_exact', 'shrink_to_fit'):
                return UNIT
            if name == 'capacity':
                return Int(len(v.a), 'usize')
            if name == 'drain':
                r = deref(args[0])
                lo = r.lo.v if r.lo is not None else 0
                hi = (r.hi.v + (1 if r.incl else 0)) if r.hi is not None else len(v.a)
                out = v.a[lo:hi]
                del v.a[lo:hi]
                return RIter(lst=out)
        # fall back to the iterator adaptors (`slice.map(..)` does not exist, but `.iter()` sugar is harmless)
        return self.iter_method(RIter(lst=a[o:o + n]), name, args, env, gargs)

    def iter_method(self, it, name, args, env, gargs):
        if name in ('iter', 'into_iter', 'by_ref', 'iter_mut', 'fuse', 'peekable'):
            return it
        if name in ('remainder', 'into_remainder'):
            return it.rem
        if name == 'rev':
            return RIter(lst=it.tolist()[::-1])
        if name == 'enumerate':
            if it.lst is not None:
                return RIter(lst=[(Int(i, 'usize'), x) for i, x in enumerate(it.lst)])
            return RIter((((Int(i, 'usize'), x)) for i, x in enumerate(it)))
        if name == 'zip':
            other = self.into_iter(args[0])
            if it.lst is not None and other.lst is not None:
                return RIter(lst=list(zip(it.lst, other.lst)))
            return RIter(zip(iter(it), iter(other)))
        if name == 'chain':
            other = self.into_iter(args[0])
            return RIter(lst=it.tolist() + other.tolist())
        if name == 'step_by':
            k = deref(args[0]).v
            if it.lst is not None and not isinstance(it.lst, IterList):
                return RIter(lst=it.lst[::k])

            def g():
                for i, x in enumerate(it):
                    if i % k == 0:
                        yield x
            return RIter(g())
        if name == 'take':
            k = deref(args[0]).v
            if it.lst is not None and not isinstance(it.lst, IterList):
                return RIter(lst=it.lst[:k])

            def g():
                if k == 0:
                    return
                for i, x in enumerate(it):
                    yield x
                    if i + 1 >= k:
                        break
            return RIter(g())
        if name == 'skip':
            k = deref(args[0]).v
            if it.lst is not None and not isinstance(it.lst, IterList):
                return RIter(lst=it.lst[k:])

            def g():
                for i, x in enumerate(it):
                    if i >= k:
                        yield x
            return RIter(g())
        if name in ('copied', 'cloned'):
            src = it.lst if it.lst is not None else it
            return RIter(lst=[copyval(deref(x)) for x in src])
        if name == 'map':
            f = args[0]
            return RIter((self.call_value(f, [x]) for x in it))
        if name == 'for_each':
            for x in it:
                self.call_value(args[0], [x])
            return UNIT
        if name == 'filter':
            f = args[0]
            return RIter((x for x in it if truth(self.call_value(f, [x]))))
        if name == 'filter_map':
            def g():
                for x in it:
                    r = self.call_value(args[0], [x])
                    if r.variant == 'Some':
                        yield r.f['0']
            return RIter(g())
        if name == 'flat_map':
            def g():
                for x in it:
                    for y in self.into_iter(self.call_value(args[0], [x])):
                        yield y
            return RIter(g())
        if name == 'flatten':
            def g():
                for x in it:
                    for y in self.into_iter(x):
                        yield y
            return RIter(g())
        if name == 'take_while':
            def g():
                for x in it:
                    if not truth(self.call_value(args[0], [x])):
                        break
                    yield x
            return RIter(g())
        if name == 'skip_while':
            def g():
                skipping = True
                for x in it:
                    if skipping and truth(self.call_value(args[0], [x])):
                        continue
                    skipping = False
                    yield x
            return RIter(g())
        if name in ('sum', 'product'):
            acc = None
            for x in it:
                x = deref(x)
                acc = x if acc is None else self.binop('+' if name == 'sum' else '*', acc, x)
            if acc is None:
                acc = Int(0 if name == 'sum' else 1)
            if gargs and gargs[0][0] == 'gtype':
                acc = self.coerce(acc, gargs[0][1], env or Env())
            return acc
        if name == 'fold':
            acc = args[0]
            for x in it:
                acc = self.call_value(args[1], [acc, x])
            return acc
        if name == 'count':
            return Int(len(it.tolist()), 'usize')
        if name == 'len':
            return Int(len(it.tolist()), 'usize')
        if name == 'last':
            l = it.tolist()
            return some(l[-1]) if l else NONE
        if name == 'next':
            if it.lst is not None:
                if isinstance(it.lst, IterList):
                    return it.lst.next()
                if not it.lst:
                    return NONE
                return some(it.lst.pop(0))
            try:
                return some(next(it.it))
            except StopIteration:
                return NONE
        if name == 'next_back':
            l = it.tolist()
            return some(l.pop()) if l else NONE
        if name == 'nth':
            l = it.tolist()
            k = deref(args[0]).v
            if k < len(l):
                x = l[k]
                del l[:k + 1]
                return some(x)
            l.clear()
            return NONE
        if name == 'collect':
            out = []
            for x in it:  # an iterator of `Result`s collects into `Result<Vec<_>, E>` (the only use in the texts run here): the
                xv = deref(x)  # first Err ends it, as FromIterator for Result does
                if isinstance(xv, Enum) and xv.enum == 'Result':
                    if xv.variant == 'Err':
                        return xv
                    out.append(('ok', xv.f['0']))
                else:
                    out.append(('v', x))
            if out and all(k == 'ok' for k, _ in out):
                return ok(Arr([v for _, v in out], True))
            return Arr([x if k == 'v' else ok(x) for k, x in out], True)
        if name in ('min', 'max'):
            l = [deref(x) for x in it]
            if not l:
                return NONE
            return some((min if name == 'min' else max)(l, key=sort_key))
        if name in ('min_by_key', 'max_by_key'):
            l = list(it)
            if not l:
                return NONE
            return some((min if name[:3] == 'min' else max)(l, key=lambda x: sort_key(self.call_value(args[0], [x]))))
        if name in ('all', 'any'):
            f = args[0]
            if name == 'all':
                return all(truth(self.call_value(f, [x])) for x in it)
            return any(truth(self.call_value(f, [x])) for x in it)
        if name in ('position', 'rposition'):
            l = it.tolist()
            idxs = range(len(l)) if name == 'position' else range(len(l) - 1, -1, -1)
            for i in idxs:
                if truth(self.call_value(args[0], [l[i]])):
                    return some(Int(i, 'usize'))
            return NONE
        if name == 'find':
            for x in it:
                if truth(self.call_value(args[0], [x])):
                    return some(x)
            return NONE
        if name == 'find_map':
            for x in it:
                r = self.call_value(args[0], [x])
                if r.variant == 'Some':
                    return r
            return NONE
        if name == 'unzip':
            l = it.tolist()
            return (Arr([x[0] for x in l], True), Arr([x[1] for x in l], True))
        if name == 'cycle':
            l = it.tolist()

            def g():
                while True:
                    for x in l:
                        yield x
            return RIter(g())
        if name == 'scan':
            st = TempPlace(args[0])

            def g():
                for x in it:
                    r = self.call_value(args[1], [st, x])
                    if r.variant == 'None':
                        break
                    yield r.f['0']
            return RIter(g())
        if name == 'inspect':
            return it
        raise InterpError('no iterator method %s' % name)

    def enum_method(self, v, name, args, env):
        if v.enum in ('Option', 'Result'):
            good = v.variant in ('Some', 'Ok')
            if name in ('unwrap', 'expect'):
                if good:
                    return v.f['0']
                raise RustPanic('called `%s::%s()` on a `%s` value' % (v.enum, name, v.variant))
            if name in ('is_some', 'is_ok'):
                return good
            if name in ('is_none', 'is_err'):
                return not good
            if name == 'unwrap_or':
                return v.f['0'] if good else args[0]
            if name == 'unwrap_or_default':
                return v.f['0'] if good else UNINIT
            if name == 'unwrap_or_else':
                return v.f['0'] if good else self.call_value(args[0], [] if v.enum == 'Option' else [v.f['0']])
            if name == 'map':
                return Enum(v.enum, v.variant, {'0': self.call_value(args[0], [v.f['0']])}) if good else v
            if name == 'map_or':
                return self.call_value(args[1], [v.f['0']]) if good else args[0]
            if name == 'map_err':
                return v if good else Enum(v.enum, v.variant, {'0': self.call_value(args[0], [v.f['0']])})
            if name == 'and_then':
                return self.call_value(args[0], [v.f['0']]) if good else v
            if name == 'or_else':
                return v if good else self.call_value(args[0], [] if v.enum == 'Option' else [v.f['0']])
            if name == 'or':
                return v if good else args[0]
            if name == 'ok_or':
                return ok(v.f['0']) if good else err(args[0])
            if name == 'ok_or_else':
                return ok(v.f['0']) if good else err(self.call_value(args[0], []))
            if name == 'ok':
                return some(v.f['0']) if good else NONE
            if name == 'err':
                return NONE if good else some(v.f['0'])
            if name in ('as_ref', 'as_mut', 'as_deref', 'as_deref_mut', 'copied', 'cloned', 'clone', 'iter', 'take') and name != 'take':
                return v if name != 'iter' else self.into_iter(v)
            if name in ('filter',):
                return v if good and truth(self.call_value(args[0], [v.f['0']])) else NONE
            if name == 'unwrap_unchecked':
                return v.f['0']
            if name in ('get_or_insert_with', 'get_or_insert') and v.enum == 'Option':
                if not good:
                    if v is NONE:
                        raise InterpError('Option::%s on a temporary' % name)
                    v.variant, v.f = 'Some', {'0': self.call_value(args[0], []) if name == 'get_or_insert_with' else args[0]}
                return v.f['0']
        if name == 'clone':
            return copyval(v)
        if name in ('eq', 'ne'):
            r = values_equal(v, args[0])
            return r if name == 'eq' else not r
        if name == 'take' and v.enum == 'Option':  # leaves None in place, returns what was there
            old = Enum('Option', v.variant, dict(v.f) if v.f else None)
            v.variant, v.f = 'None', None
            return old
        raise InterpError('no method %s on %r' % (name, v))


class TryInto:
    __slots__ = ('v',)

    def __init__(self, v):
        self.v = v


class LazyLocal:
    __slots__ = ('lz',)

    def __init__(self, lz):
        self.lz = lz


class IterList:
    """An endless range (a..) used as a list-like source."""

    def __init__(self, gen):
        self.gen = gen

    def __iter__(self):
        return self.gen

    def next(self):
        return some(next(self.gen))


def deref_once(v):
    if isinstance(v, Place):
        return v.get()
    return v


def sort_key(v):
    v = deref(v)
    if isinstance(v, Int):
        return v.v
    if isinstance(v, tuple):
        return tuple(sort_key(x) for x in v)
    if isinstance(v, Enum):
        return (v.variant,)
    return v


# --------------------------------------------------------------------------------------------- numpy bridges

def _depth(v):
    lst = seq_list(v)
    if lst and isinstance(deref(lst[0]), (Arr, Slice)):
        return 1 + _depth(lst[0])
    return 1


def f32_array(values):
    """numpy 1-D (one lane) or 2-D [n, lanes] float32 -> Arr of f32 scalars / lane vectors."""
    v = np.asarray(values, dtype=np.float32)
    if v.ndim == 1:
        return Arr([F32(x) for x in v])
    return Arr([np.array(row, dtype=np.float32) for row in v])


def int_array(values, ty):
    return Arr([Int(int(x), ty) for x in np.asarray(values).ravel()])


def to_numpy(v, lanes=None):
    """Interpreter array -> numpy.  `lanes`: the batch width, when the run was batched (rows that hold only scalars, e.g.
    after `fill(0.0)`, are then broadcast to it)."""
    if lanes is not None:
        r = to_numpy(v)
        if r.dtype == np.float32:
            want_nd = 1 + _depth(v)
            if r.ndim < want_nd:
                r = np.broadcast_to(r[..., None], r.shape + (lanes,)).copy()
        return r
    lst = seq_list(v)
    if not lst:
        return np.zeros(0, np.float32)
    first = deref(lst[0])
    if isinstance(first, Int):
        return np.array([deref(x).v for x in lst], dtype=np.int64)
    if isinstance(first, (Arr, Slice)):
        rows = [to_numpy(x) for x in lst]
        nd = max(r.ndim for r in rows)
        if any(r.ndim < nd for r in rows):
            shape = next(r.shape for r in rows if r.ndim == nd)
            rows = [r if r.ndim == nd else np.broadcast_to(r[..., None], shape).copy() for r in rows]
        return np.stack(rows)
    if isinstance(first, Struct) and set(first.f) == {'re', 'im'}:
        return np.array([[deref(x).f['re'], deref(x).f['im']] for x in lst])
    out = []
    for x in lst:
        x = deref(x)
        if isinstance(x, ULit):
            x = x.resolve('f32')
        out.append(x)
    if isinstance(out[0], float):
        return np.array(out, dtype=np.float64)
    lanes = [x.shape for x in out if isinstance(x, np.ndarray) and x.ndim]
    if lanes:  # a batch: scalars (literals stored into a lane vector's place) stand for every lane
        return np.stack([np.broadcast_to(np.asarray(x, dtype=np.float32), lanes[0]) for x in out])
    return np.array(out, dtype=np.float32)
