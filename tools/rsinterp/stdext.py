"""std::sync and std::collections for the interpreter: what the trait-side shim (bindings/rust/symphonia-accel-hip) needs to
be EXECUTED by tests/test_rust_shim.py -- Arc / Weak / Mutex / OnceLock, HashMap with the entry API, VecDeque, #[derive(Default)].
The DSP fixtures (tools/rs2fixtures.py) use none of it.

  Arc<T> / Mutex<T>   `Cell` wrappers with reference identity: `clone()` of an Arc is the same Arc, `lock()` yields the inner
                      object itself (Ok(..)), every other method and every field access goes through to the inner value;
  Weak<T>             a Cell around the Arc: `upgrade()` is always Some (nothing is ever dropped here), strong_count 1;
  HashMap<K, V>       `HMap`: a dict keyed by a hashable image of the Rust key, values by reference; the value type (from the
                      declared field type) is remembered so that `entry(k).or_default()` can build one;
  VecDeque<T>         an `Arr` with the deque methods (interp.seq_method);
  T::default()        for structs: field-wise from the declared types.
"""
from . import interp as I


class Cell:
    __slots__ = ('kind', 'v')

    def __init__(self, kind, v):
        self.kind, self.v = kind, v

    def __repr__(self):
        return '%s(%r)' % (self.kind, self.v)


class HMap:
    __slots__ = ('d', 'vty', 'ordered', 'is_set')

    def __init__(self, vty=None, ordered=False, is_set=False):
        # image of key -> (key, value); ordered: a BTreeMap (iterates by key); is_set: a HashSet (the values are unit)
        self.d, self.vty, self.ordered, self.is_set = {}, vty, ordered, is_set

    def items(self):
        return [self.d[h] for h in sorted(self.d)] if self.ordered else list(self.d.values())

    def __repr__(self):
        return 'HashMap(%d)' % len(self.d)


class HEntry:
    __slots__ = ('m', 'k')

    def __init__(self, m, k):
        self.m, self.k = m, k


def hkey(v):
    v = I.deref(v)
    if isinstance(v, I.Int):
        return ('i', v.v)
    if isinstance(v, I.Struct):
        return (v.name,) + tuple(hkey(x) for x in v.f.values())
    if isinstance(v, I.Enum):
        return (v.enum, v.variant) + tuple(hkey(x) for x in (v.f or {}).values())
    if isinstance(v, tuple):
        return tuple(hkey(x) for x in v)
    if isinstance(v, (bool, str)):
        return v
    raise I.InterpError('unhashable HashMap key %r' % (v,))


def inner(v):
    """through any number of Arc / Mutex layers"""
    while isinstance(v, Cell) and v.kind in ('Arc', 'Mutex'):
        v = v.v
    return v


def default_of(it, ty, env=None):
    if ty is None:
        return I.UNINIT
    k = ty[0]
    if k == 'tarray':  # [T; N]: N defaults (Default is implemented for arrays of up to 32 elements)
        n = I.deref(it.ev(ty[2], env or I.Env()))
        first = default_of(it, ty[1], env)
        if first is I.UNINIT:
            return I.UNINIT
        return I.Arr([first] + [default_of(it, ty[1], env) for _ in range(n.v - 1)])
    if k == 'tpath':
        name, gargs = ty[1][-1], ty[2]
        if name in I.INT_BITS:
            return I.Int(0, name)
        if name == 'bool':
            return False
        if name == 'f32':
            return I.F32(0.0)
        if name == 'f64':
            return 0.0
        if name == 'Option':
            return I.Enum('Option', 'None')
        if name in ('Vec', 'VecDeque'):
            return I.Arr([], True)
        if name in ('HashMap', 'BTreeMap'):
            vty = gargs[1][1] if len(gargs) >= 2 and gargs[1][0] == 'gtype' else None
            return HMap(vty, name == 'BTreeMap')
        if name in it.types and 'default' in it.impls.get(name, {}):  # a hand-written `impl Default`
            m = it.impls[name]['default']
            if isinstance(m, tuple) and m[0] == 'fn':
                return it.call_fn(m, [], None, name, None)
        if name in it.types and it.types[name][0] == 'struct':
            item = it.types[name]
            return I.Struct(name, {f: default_of(it, t, env) for f, t in item[3]})
    if k == 'ttuple':
        return tuple(default_of(it, t, env) for t in ty[1])
    return I.UNINIT


def path_builtin(it, segs):
    name = segs[-1]
    head = segs[-2] if len(segs) >= 2 else None
    from . import ffi
    raw = ffi.path_builtin(segs)
    if raw is not None:
        return raw
    # std::io::Error: modelled as the symphonia Error it becomes at the first `?` (errors.rs `impl From<io::Error> for Error`)
    if head == 'ErrorKind' and 'io' in segs:
        return name
    if head == 'Error' and 'io' in segs and name in ('other', 'new', 'from'):
        return I.Builtin(lambda *a: I.Enum('Error', 'IoError', {'0': a[-1] if a else ''}), 'io::Error::' + name)
    if head in ('Arc', 'Rc') and name == 'new':
        return I.Builtin(lambda v: Cell('Arc', v), 'Arc::new')
    if head in ('Arc', 'Rc') and name == 'downgrade':
        return I.Builtin(lambda a: Cell('Weak', I.deref(a)), 'Arc::downgrade')
    if head in ('Arc', 'Rc') and name == 'clone':
        return I.Builtin(lambda a: I.deref(a), 'Arc::clone')
    if head == 'Mutex' and name == 'new':
        def mutex_new(v):
            if isinstance(v, I.Enum):  # its own object: `guard.get_or_insert_with(..)` mutates it in place (never the shared None)
                v = I.Enum(v.enum, v.variant, dict(v.f) if v.f else None)
            return Cell('Mutex', v)
        return I.Builtin(mutex_new, 'Mutex::new')
    if head == 'OnceLock' and name == 'new':  # std::sync::OnceLock: an empty slot; get_or_init fills it once
        return I.Builtin(lambda: Cell('OnceLock', None), 'OnceLock::new')
    if head == 'HashMap' and name in ('new', 'with_capacity', 'default'):
        return I.Builtin(lambda *a: HMap(), 'HashMap::new')
    if head == 'HashSet' and name in ('new', 'with_capacity', 'default'):
        return I.Builtin(lambda *a: HMap(None, False, True), 'HashSet::new')
    if head == 'BTreeMap' and name in ('new', 'default'):
        return I.Builtin(lambda *a: HMap(None, True), 'BTreeMap::new')
    if head == 'NonZero' and name == 'new':  # NonZero<T> is its integer; `get()` gives it back (interp.int_method)
        return I.Builtin(lambda v: I.some(I.deref(v)) if I.deref(v).v != 0 else I.NONE, 'NonZero::new')
    if head == 'VecDeque' and name in ('new', 'with_capacity', 'default'):
        return I.Builtin(lambda *a: I.Arr([], True), 'VecDeque::new')
    if name == 'default' and head in it.types and it.types[head][0] == 'struct' and 'default' not in it.impls.get(head, {}):
        return I.Builtin(lambda _h=head: default_of(it, ('tpath', [_h], [])), head + '::default')
    return None


def is_std(v):
    return isinstance(v, (Cell, HMap, HEntry))


def method(it, base, name, args, env):
    """Returns (handled, value)."""
    if isinstance(base, Cell):
        if base.kind == 'Arc':
            if name == 'clone':
                return True, base
            return False, base.v        # auto-deref: the caller retries on the inner value
        if base.kind == 'Mutex':
            if name in ('lock', 'try_lock'):
                return True, I.ok(base.v)
            if name in ('get_mut', 'into_inner'):
                return True, I.ok(base.v)
            raise I.InterpError('no method %s on Mutex' % name)
        if base.kind == 'OnceLock':
            if name == 'get_or_init':
                if base.v is None:
                    base.v = it.call_value(args[0], [])
                return True, base.v
            if name == 'get':
                return True, (I.some(base.v) if base.v is not None else I.NONE)
            raise I.InterpError('no method %s on OnceLock' % name)
        if base.kind == 'Weak':
            if name == 'upgrade':
                return True, I.some(base.v)
            if name == 'strong_count':
                return True, I.Int(1, 'usize')
            if name == 'clone':
                return True, base
            raise I.InterpError('no method %s on Weak' % name)
    if isinstance(base, HMap):
        d = base.d
        if name in ('get', 'get_mut'):
            e = d.get(hkey(args[0]))
            return True, (I.some(e[1]) if e is not None else I.NONE)
        if name == 'contains_key' or (name == 'contains' and base.is_set):
            return True, hkey(args[0]) in d
        if name == 'insert' and base.is_set:  # HashSet::insert: true if the value was not there
            k = I.copyval(I.deref(args[0]))
            fresh = hkey(k) not in d
            d[hkey(k)] = (k, I.UNIT)
            return True, fresh
        if name == 'insert':
            k = I.copyval(I.deref(args[0]))
            old = d.get(hkey(k))
            d[hkey(k)] = (k, args[1])
            return True, (I.some(old[1]) if old is not None else I.NONE)
        if name == 'remove':
            old = d.pop(hkey(args[0]), None)
            return True, (I.some(old[1]) if old is not None else I.NONE)
        if name == 'entry':
            return True, HEntry(base, I.copyval(I.deref(args[0])))
        if name == 'len':
            return True, I.Int(len(d), 'usize')
        if name == 'is_empty':
            return True, not d
        if name == 'clear':
            d.clear()
            return True, I.UNIT
        if name in ('values', 'values_mut', 'into_values'):
            return True, I.RIter(lst=[v for _, v in base.items()])
        if name in ('keys', 'into_keys'):
            return True, I.RIter(lst=[k for k, _ in base.items()])
        if name in ('iter', 'iter_mut', 'into_iter'):
            return True, I.RIter(lst=[(k, v) for k, v in base.items()])
        if name == 'clone':
            m = HMap(base.vty, base.ordered)
            m.d = {h: (I.copyval(k), I.deepclone(v) if I.is_agg(v) else v) for h, (k, v) in d.items()}
            return True, m
        raise I.InterpError('no method %s on HashMap' % name)
    if isinstance(base, HEntry):
        m, k = base.m, base.k
        h = hkey(k)
        if name in ('or_default', 'or_insert', 'or_insert_with'):
            if h not in m.d:
                if name == 'or_default':
                    if m.vty is None:
                        raise I.InterpError('HashMap::entry().or_default(): the value type is not known here')
                    v = default_of(it, m.vty)
                elif name == 'or_insert':
                    v = args[0]
                else:
                    v = it.call_value(args[0], [])
                m.d[h] = (k, v)
            return True, m.d[h][1]
        raise I.InterpError('no method %s on hash_map::Entry' % name)
    return False, base
