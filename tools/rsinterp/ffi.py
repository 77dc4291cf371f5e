"""`unsafe extern "C"` for the interpreter: the raw-pointer idioms of bindings/rust/symphonia-accel-hip/src/ctx.rs and the calls
of its codec adapters into libsymaccel, bound through ctypes (tests/test_rust_adapters.py, tests/test_flac_packets.py).

What the crate does with raw pointers is little, and all of it is modelled by value:
  ptr::null_mut() / ptr::null()         NULL (is_null() is true)
  out-pointer to an opaque handle       `symaccel_ctx_create(device, &mut raw)`: the place receives a `Handle` (the C pointer)
  symaccel_host_alloc(bytes, &mut p)    the place receives a `RawMem`: an element list created on first use
  p as *mut T                           the same object (casts between pointer types do nothing)
  ptr.add(i).write(v)                   stores element i
  slice::from_raw_parts(_mut)(ptr, n)   a slice over the first n elements
  v.as_ptr() / v.as_mut_ptr()           the Vec / slice itself (interp.builtin_method)
  CStr::from_ptr(p)                     the Python str a `*const c_char` result was turned into
  pointers INSIDE a struct the library fills in (symaccel_batch_slot: the planes of a reservation with the cross-stream batcher)
                                        are REAL addresses: `CPtr`.  `p as *mut T` gives the address an element type,
                                        slice::from_raw_parts(_mut) over it is a slice whose storage is the C memory itself (`MemList`):
                                        what the crate writes there is what libsymaccel's copy kernels read -- the zero-copy path of
                                        ctx.rs `BatchSlot` is executed, not modelled

A call of a bound function marshals every pointer argument into a numpy array of the element type the DECLARATION in
bindings/rust/symaccel_sys.rs gives it (scalars, or #[repr(C)] structs as structured dtypes), calls the C function, and copies
`*mut` arrays back element by element.  The calls the crate makes are synchronous host-pointer entry points, so copy-in /
copy-out is exactly their contract."""
import ctypes as C
import re

import numpy as np

from . import interp as I

SCALARS = {'u8': np.uint8, 'i8': np.int8, 'u16': np.uint16, 'i16': np.int16, 'u32': np.uint32, 'i32': np.int32, 'u64': np.uint64,
           'i64': np.int64, 'usize': np.uint64, 'isize': np.int64, 'f32': np.float32, 'f64': np.float64, 'c_int': np.int32, 'c_uint': np.uint32,
           'c_char': np.int8, 'bool': np.uint8}
CT = {'u8': C.c_uint8, 'i8': C.c_int8, 'u16': C.c_uint16, 'i16': C.c_int16, 'u32': C.c_uint32, 'i32': C.c_int32, 'u64': C.c_uint64,
      'i64': C.c_int64, 'usize': C.c_size_t, 'isize': C.c_ssize_t, 'f32': C.c_float, 'f64': C.c_double, 'c_int': C.c_int, 'c_uint': C.c_uint}


class Null:
    def __repr__(self):
        return 'null'

    def rs_method(self, it, name, args):
        if name == 'is_null':
            return True
        raise I.InterpError('no method %s on a null pointer' % name)


NULL = Null()


class Handle:
    """an opaque C pointer (symaccel_ctx *, a communicator)"""

    def __init__(self, p):
        self.p = p

    def __repr__(self):
        return 'Handle(%#x)' % (self.p or 0)

    def rs_method(self, it, name, args):
        if name == 'is_null':
            return not self.p
        raise I.InterpError('no method %s on an opaque pointer' % name)


class RawMem:
    """memory from symaccel_host_alloc: `nbytes` bytes seen as elements of whatever the crate stores in it"""

    def __init__(self, nbytes):
        self.nbytes, self.arr, self.off = nbytes, I.Arr([], True), 0

    def __repr__(self):
        return 'RawMem(%d B, %d elements)' % (self.nbytes, len(self.arr.a))

    def need(self, n):
        a = self.arr.a
        if len(a) < n:
            a.extend([I.Int(0)] * (n - len(a)))

    def rs_method(self, it, name, args):
        if name == 'is_null':
            return False
        if name in ('add', 'offset'):
            view = RawMem(self.nbytes)
            view.arr, view.off = self.arr, self.off + int(I.deref(args[0]).v)
            return view
        if name == 'write':
            self.need(self.off + 1)
            self.arr.a[self.off] = args[0]
            return I.UNIT
        if name == 'read':
            self.need(self.off + 1)
            return self.arr.a[self.off]
        if name in ('cast', 'cast_mut', 'cast_const'):
            return self
        raise I.InterpError('no method %s on a raw pointer' % name)


class CPtr:
    """a real address inside memory libsymaccel owns (a plane of a symaccel_batch_slot), with the element type the crate cast it to"""

    def __init__(self, bridge, addr, elem='c_void'):
        self.bridge, self.addr, self.elem = bridge, int(addr), elem

    def __repr__(self):
        return 'CPtr(%#x as *%s)' % (self.addr, self.elem)

    def rs_ptr_cast(self, elem):
        return CPtr(self.bridge, self.addr, elem)

    def rs_method(self, it, name, args):
        if name == 'is_null':
            return self.addr == 0
        if name in ('add', 'offset'):
            return CPtr(self.bridge, self.addr + int(I.deref(args[0]).v) * self.bridge.dtype(self.elem).itemsize, self.elem)
        if name in ('cast', 'cast_mut', 'cast_const'):
            return self
        if name in ('read', 'write'):
            mem = MemList(self.bridge, self.addr, self.elem, 1)
            if name == 'read':
                return mem[0]
            mem[0] = args[0]
            return I.UNIT
        raise I.InterpError('no method %s on a raw pointer' % name)


class MemList:
    """list-like storage of a Slice over C memory: element reads and writes go to the memory itself"""

    def __init__(self, bridge, addr, elem, n):
        self.bridge, self.elem, self.n = bridge, elem, n
        self.dt = bridge.dtype(elem)
        self.like = I.Struct(elem.split('::')[-1], {}) if self.dt.names else None
        self.arr = np.frombuffer((C.c_char * (n * self.dt.itemsize)).from_address(addr), dtype=self.dt) if n else np.zeros(0, self.dt)

    def __len__(self):
        return self.n

    def _get(self, i):
        return self.bridge.from_np(self.arr[i], self.dt, self.like)

    def _set(self, i, v):
        self.arr[i] = self.bridge.to_np(I.deref(v), self.dt)

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self._get(j) for j in range(*i.indices(self.n))]
        if i < 0 or i >= self.n:
            raise I.RustPanic('index out of bounds: the len is %d but the index is %d' % (self.n, i))
        return self._get(i)

    def __setitem__(self, i, v):
        if isinstance(i, slice):
            idx = range(*i.indices(self.n))
            v = list(v)
            if len(v) != len(idx):
                raise I.InterpError('a slice over C memory cannot change its length')
            if v and not self.dt.names and all(isinstance(x, (I.Int, float, np.floating, int)) for x in v):  # the common case in one go
                self.arr[idx.start:idx.stop:idx.step] = np.array([x.v if isinstance(x, I.Int) else x for x in v]).astype(self.dt)
                return
            for j, x in zip(idx, v):
                self._set(j, x)
            return
        if i < 0 or i >= self.n:
            raise I.RustPanic('index out of bounds: the len is %d but the index is %d' % (self.n, i))
        self._set(i, v)

    def __iter__(self):
        return (self._get(i) for i in range(self.n))


def from_raw_parts(ptr, n, mut):
    ptr = I.deref(ptr)
    n = int(I.deref(n).v)
    if isinstance(ptr, CPtr):
        if ptr.elem.split('::')[-1] == 'c_void':
            raise I.InterpError('slice::from_raw_parts over an untyped pointer: cast it first')
        if ptr.addr == 0 and n:
            raise I.RustPanic('slice::from_raw_parts over a null pointer')
        return I.Slice(MemList(ptr.bridge, ptr.addr, ptr.elem, n), 0, n, mut)
    if isinstance(ptr, RawMem):
        ptr.need(ptr.off + n)
        return I.Slice(ptr.arr.a, ptr.off, n, mut)
    if isinstance(ptr, (I.Arr, I.Slice)):
        a, o, _ = I.seq_view(ptr)
        return I.Slice(a, o, n, mut)
    raise I.InterpError('slice::from_raw_parts of %r' % (ptr,))


def path_builtin(segs):
    name, head = segs[-1], (segs[-2] if len(segs) >= 2 else None)
    if head == 'ptr' and name in ('null', 'null_mut'):
        return I.Builtin(lambda: NULL, 'ptr::' + name)
    if head == 'slice' and name in ('from_raw_parts', 'from_raw_parts_mut'):
        return I.Builtin(lambda p, n, _m=name.endswith('mut'): from_raw_parts(p, n, _m), 'slice::' + name)
    if head == 'CStr' and name == 'from_ptr':
        return I.Builtin(lambda p: CStrVal(p if isinstance(p, str) else ''), 'CStr::from_ptr')
    return None


class CStrVal:
    def __init__(self, s):
        self.s = s

    def rs_method(self, it, name, args):
        if name == 'to_str':
            return I.ok(self.s)
        if name in ('to_string_lossy', 'into_owned', 'to_owned', 'to_string'):
            return self.s
        raise I.InterpError('no method %s on CStr' % name)


# ---------------------------------------------------------------------------------------------- declarations

def parse_extern_block(text):
    """(text without the extern block, {fn name: ([(param, type text)], return type text or None)})"""
    m = re.search(r'extern\s+"C"\s*\{', text)
    if not m:
        return text, {}
    depth, i = 1, m.end()
    while depth:
        depth += {'{': 1, '}': -1}.get(text[i], 0)
        i += 1
    block = text[m.end():i - 1]
    decls = {}
    for fm in re.finditer(r'pub fn (\w+)\(([^)]*)\)\s*(?:->\s*([^;]+))?;', block):
        params = []
        for prm in [x for x in fm.group(2).split(',') if x.strip()]:
            pn, pt = prm.split(':', 1)
            params.append((pn.strip(), ' '.join(pt.split())))
        decls[fm.group(1)] = (params, fm.group(3).strip() if fm.group(3) else None)
    head = re.sub(r'(#\[link[^\]]*\]\s*)?(unsafe)?\s*$', '', text[:m.start()])
    return head + '\n' + text[i:], decls


class Bridge:
    """binds the `extern "C"` declarations of a bindings file to a ctypes library inside one interpreter"""

    def __init__(self, it, sys_rs_text, dll):
        # a CDLL object of its own on the same loaded library: call() sets restype / argtypes per function, and ctypes keeps those on
        # the CDLL's function objects -- the caller's handle (symphonia_amd._ffi, which declares argtypes) must not see them change
        self.it, self.dll = it, C.CDLL(dll._name)
        rest, self.decls = parse_extern_block(sys_rs_text)
        it.load_source(rest, 'symaccel_sys.rs')
        # symaccel_batcher_submit_* return before the library writes the `*_io` / `pcm` arrays (symaccel_batcher_collect does): the
        # marshalled arrays of a submission are kept, keyed by its ticket, and copied back into the interpreter's values at collect
        self.deferred = {}
        self.ptr_fields = set()  # (struct, field) pairs that hold addresses
        self.word_fields = {}    # (struct, field) -> 'usize' / 'isize'
        it.size_of_struct = lambda name: self.dtype(name).itemsize  # std::mem::size_of::<T>() of a #[repr(C)] record
        self.calls = []  # (name) log, for the tests
        self.scalars = []  # per call: (name, {parameter: value}) for the integer arguments passed by value
        for name in self.decls:
            it.globals[name] = I.Builtin(lambda *a, _n=name: self.call(_n, *a), name)
            it.globals['ffi::' + name] = it.globals[name]

    # -- element types
    def dtype(self, ty):
        ty = ty.strip()
        base = ty.split('::')[-1]
        if base in SCALARS:
            return np.dtype(SCALARS[base])
        item = self.it.types.get(base)
        if item is None or item[0] != 'struct':
            raise I.InterpError('ffi: no layout for %s' % ty)
        fields = []
        for fname, fty in item[3]:
            fields.append((fname,) + self.field_dtype(fty))
            inner = fty
            while inner[0] == 'tarray':
                inner = inner[1]
            if inner[0] == 'tptr':
                self.ptr_fields.add((base, fname))
            elif inner[0] == 'tpath' and inner[1][-1] in ('usize', 'isize'):
                self.word_fields[(base, fname)] = inner[1][-1]  # (u64 / i64 on the wire; `usize` to the crate)
        return np.dtype(fields)

    def field_dtype(self, fty):
        if fty[0] == 'tptr':  # an address (symaccel_batch_slot's planes)
            return (np.dtype(np.uint64),)
        if fty[0] == 'tarray':
            n = int(self.it.ev(fty[2], I.Env()).v) if not isinstance(fty[2], int) else fty[2]
            inner = self.field_dtype(fty[1])
            return (inner[0], (n,) + (inner[1] if len(inner) > 1 else ()))
        return (self.dtype(self.it.type_name(fty)),)

    def to_np(self, v, dt):
        if isinstance(v, I.Struct):
            out = []
            for fname in dt.names:
                sub = dt.fields[fname][0]
                x = v.f[fname]
                out.append(self.to_np_seq(x, sub.base, sub.shape) if sub.shape else self.to_np(x, sub))
            return tuple(out)
        if isinstance(v, I.Int):
            return v.v
        if isinstance(v, CPtr):
            return v.addr
        if isinstance(v, (Null, Handle)):
            return getattr(v, 'p', 0) or 0
        if v is I.UNINIT or v is None:
            return 0
        if isinstance(v, (bool, np.bool_)):
            return int(v)
        if isinstance(v, I.ULit):  # an untyped float literal that never met a typed operand (`vec.resize(n, 0.0)`): the array gives the type
            return v.resolve('f64' if dt == np.float64 else 'f32')
        return v

    def to_np_seq(self, v, dt, shape=None):
        a, o, n = I.seq_view(I.deref(v))
        if dt.names:
            arr = np.array([self.to_np(x, dt) for x in a[o:o + n]], dtype=dt)
        else:
            arr = np.array([self.to_np(x, dt) for x in a[o:o + n]]).astype(dt) if n else np.zeros(0, dt)
        return arr

    def from_np(self, x, dt, like):
        if dt.names:
            f = {}
            sname = like.name if isinstance(like, I.Struct) else '?'
            for fname in dt.names:
                sub = dt.fields[fname][0]
                ptr = (sname, fname) in self.ptr_fields
                word = self.word_fields.get((sname, fname))
                if ptr:
                    conv = lambda y: CPtr(self, int(y)) if int(y) else NULL  # noqa: E731
                elif word:
                    conv = lambda y, _w=word: I.Int(int(y), _w)  # noqa: E731
                else:
                    conv = lambda y, _b=sub.base if sub.shape else sub: self.from_np(y, _b, None)  # noqa: E731
                if sub.shape:
                    f[fname] = I.Arr([conv(y) for y in np.asarray(x[fname]).reshape(-1)])
                else:
                    f[fname] = conv(x[fname])
            return I.Struct(sname, f)
        if dt.kind == 'f':
            return I.F32(x) if dt.itemsize == 4 else float(x)
        name = {v: k for k, v in SCALARS.items() if k[0] in 'iu' and k not in ('usize', 'isize')}.get(dt.type, 'i32')
        return I.Int(int(x), name)

    # -- one call
    def call(self, name, *args):
        params, ret = self.decls[name]
        fn = getattr(self.dll, name)
        cargs, after, keep = [], [], []
        if len(args) != len(params):
            raise I.InterpError('ffi: %s takes %d arguments, got %d' % (name, len(params), len(args)))
        for (pn, pt), v in zip(params, args):
            m = re.fullmatch(r'\*(const|mut) (.+)', pt)
            if not m:
                base = pt.split('::')[-1]
                x = I.deref(v)
                if base in ('f32', 'f64'):
                    cargs.append(CT[base](float(x)))
                else:
                    cargs.append(CT[base](int(x.v) if isinstance(x, I.Int) else int(x)))
                continue
            mut, elem = m.group(1) == 'mut', m.group(2)
            m2 = re.fullmatch(r'\*(const|mut) (.+)', elem)
            if m2:  # pointer to pointer: an out-parameter for a handle or for host memory
                if not isinstance(v, I.Place):
                    raise I.InterpError('ffi: %s: %s must be `&mut` of a pointer variable' % (name, pn))
                slot = C.c_void_p()
                cargs.append(C.byref(slot))
                after.append(('handle', v, slot, m2.group(2), args))
                continue
            x = I.deref(v)
            if isinstance(x, Null):
                cargs.append(None)
            elif isinstance(x, Handle):
                cargs.append(C.c_void_p(x.p))
            elif isinstance(x, RawMem) and elem.split('::')[-1] == 'c_void':
                cargs.append(None)  # symaccel_host_free(ptr): the emulated allocation has no C counterpart
            elif isinstance(v, I.Place) and not isinstance(x, (I.Arr, I.Slice, RawMem)):  # `&mut scalar`
                dt = self.dtype(elem)
                arr = np.array([self.to_np(x, dt)], dtype=dt)
                keep.append(arr)
                cargs.append(arr.ctypes.data_as(C.c_void_p))
                if mut:
                    after.append(('scalar', v, arr, dt, x))
            elif isinstance(x, I.Struct):  # `&record` for a `*const Record` parameter
                dt = self.dtype(elem)
                arr = np.array([self.to_np(x, dt)], dtype=dt)
                keep.append(arr)
                cargs.append(arr.ctypes.data_as(C.c_void_p))
            else:
                if isinstance(x, RawMem):
                    x = I.Slice(x.arr.a, x.off, len(x.arr.a) - x.off, True)
                dt = self.dtype(elem)
                arr = np.ascontiguousarray(self.to_np_seq(x, dt))
                keep.append(arr)
                cargs.append(arr.ctypes.data_as(C.c_void_p) if arr.size else None)
                if mut:
                    after.append(('seq', x, arr, dt, None))
        rbase = ret.split('::')[-1] if ret else None
        if ret is None:
            fn.restype = None
        elif ret.startswith('*'):
            fn.restype = C.c_char_p if 'c_char' in ret else C.c_void_p
        else:
            fn.restype = CT[rbase]
        fn.argtypes = None
        self.calls.append(name)
        self.scalars.append((name, {pn: int(c.value) for (pn, pt), c in zip(params, cargs) if hasattr(c, 'value') and isinstance(c.value, int) and not pt.startswith('*')}))
        r = fn(*cargs)
        if name.startswith('symaccel_batcher_submit_') and int(r) == 0:
            ticket = [arr for kind, dst, arr, dt, extra in after if kind == 'scalar'][-1]
            self.deferred[int(ticket[0])] = ([a for a in after if a[0] == 'seq'], keep)
            after = [a for a in after if a[0] != 'seq']
        if name in ('symaccel_batcher_collect', 'symaccel_batcher_abandon'):
            held = self.deferred.pop(int(cargs[1].value), None)
            if held and name.endswith('collect') and int(r) == 0:
                after = after + held[0]
        for kind, dst, arr, dt, extra in after:
            if kind == 'seq':
                a, o, n = I.seq_view(dst)
                for i in range(n):
                    a[o + i] = self.from_np(arr[i], dt, a[o + i])
            elif kind == 'scalar':
                dst.set(self.from_np(arr[0], dt, extra))
            else:  # an out-pointer
                if 'c_void' in dt and name.endswith('host_alloc'):
                    nbytes = int(I.deref(extra[0]).v)
                    if arr.value:
                        self.dll.symaccel_host_free(C.c_void_p(arr.value))  # the crate's buffer lives in the interpreter
                    dst.set(RawMem(nbytes) if arr.value else NULL)
                else:
                    dst.set(Handle(arr.value) if arr.value else NULL)
        if ret is None:
            return I.UNIT
        if ret.startswith('*'):
            return r.decode() if isinstance(r, bytes) else (Handle(r) if r else NULL)
        if rbase in ('f32', 'f64'):
            return I.F32(r) if rbase == 'f32' else float(r)
        return I.Int(int(r), 'i32' if rbase == 'c_int' else ('u32' if rbase == 'c_uint' else rbase))
