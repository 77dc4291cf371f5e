"""Lexer and recursive-descent parser for the Rust subset (see __init__.py).  AST nodes are tuples: (tag, ...)."""
import re

TOKEN_RE = re.compile(r'''
 (?P<ws>\s+)
|(?P<lcomment>//[^\n]*)
|(?P<bcomment>/\*.*?\*/)
|(?P<rawstr>b?r(?P<h>\#*)".*?"(?P=h))
|(?P<str>b?"(?:\\.|[^"\\])*")
|(?P<char>b?'(?:\\(?:x[0-9a-fA-F]{2}|u\{[0-9a-fA-F]+\}|.)|[^'\\])')
|(?P<lifetime>'[A-Za-z_]\w*)
|(?P<float>\d[\d_]*\.\d[\d_]*(?:[eE][+-]?\d[\d_]*)?(?:f32|f64)?|\d[\d_]*[eE][+-]?\d[\d_]*(?:f32|f64)?|\d[\d_]*(?:f32|f64)|\d[\d_]*\.(?![\.\w]))
|(?P<int>0x[0-9a-fA-F_]+(?:[iu](?:8|16|32|64|128|size))?|0b[01_]+(?:[iu](?:8|16|32|64|128|size))?|0o[0-7_]+(?:[iu](?:8|16|32|64|128|size))?|\d[\d_]*(?:[iu](?:8|16|32|64|128|size))?)
|(?P<ident>r\#[A-Za-z_]\w*|[A-Za-z_]\w*)
|(?P<punct><<=|>>=|\.\.\.|\.\.=|::|->|=>|==|!=|<=|>=|&&|\|\||\+=|-=|\*=|/=|%=|\^=|&=|\|=|<<|>>|\.\.|[-+*/%^!&|=<>@.,;:\#$?~\[\](){}])
''', re.X | re.S)

INT_SUFFIX = re.compile(r'([iu](?:8|16|32|64|128|size))$')


class Tok:
    __slots__ = ('k', 's', 'line')

    def __init__(self, k, s, line):
        self.k, self.s, self.line = k, s, line

    def __repr__(self):
        return '%s:%r@%d' % (self.k, self.s, self.line)


class ParseError(Exception):
    pass


def lex(src):
    toks, pos, line, n = [], 0, 1, len(src)
    while pos < n:
        m = TOKEN_RE.match(src, pos)
        if not m:
            raise ParseError('cannot lex at line %d: %r' % (line, src[pos:pos + 30]))
        k = m.lastgroup
        s = m.group(k)
        if k == 'h':  # (inner group of rawstr)
            k, s = 'rawstr', m.group('rawstr')
        if k not in ('ws', 'lcomment', 'bcomment'):
            if k == 'rawstr':
                k = 'str'
            toks.append(Tok(k, s, line))
        line += m.group(0).count('\n')
        pos = m.end()
    toks.append(Tok('eof', '', line))
    return toks


ASSIGN_OPS = {'=', '+=', '-=', '*=', '/=', '%=', '^=', '&=', '|=', '<<=', '>>='}
BIN_PREC = {
    '*': 11, '/': 11, '%': 11, '+': 10, '-': 10, '<<': 9, '>>': 9, '&': 8, '^': 7, '|': 6,
    '==': 5, '!=': 5, '<': 5, '>': 5, '<=': 5, '>=': 5, '&&': 4, '||': 3,
}
BLOCKLIKE = {'if', 'match', 'loop', 'while', 'for', 'unsafe'}
KEYWORDS_NOT_EXPR_START = {'let', 'fn', 'const', 'static', 'struct', 'enum', 'impl', 'trait', 'use', 'mod', 'type', 'pub', 'extern'}


class Parser:
    def __init__(self, toks, fname='<src>'):
        self.t, self.i, self.fname = toks, 0, fname
        self.uses = {}

    # ------------------------------------------------------------------ token helpers
    @property
    def cur(self):
        return self.t[self.i]

    def peek(self, k=1):
        j = self.i + k
        return self.t[j] if j < len(self.t) else self.t[-1]

    def at(self, s):
        c = self.t[self.i]
        return c.s == s and c.k in ('punct', 'ident')

    def at_ident(self, s=None):
        c = self.t[self.i]
        return c.k == 'ident' and (s is None or c.s == s)

    def eat(self, s):
        if self.at(s):
            self.i += 1
            return True
        return False

    def expect(self, s):
        if not self.eat(s):
            self.err('expected %r' % s)

    def err(self, msg):
        c = self.cur
        ctx = ' '.join(x.s for x in self.t[max(0, self.i - 6):self.i + 6])
        raise ParseError('%s:%d: %s, got %r  (near: %s)' % (self.fname, c.line, msg, c.s, ctx))

    def ident(self):
        c = self.cur
        if c.k != 'ident':
            self.err('expected identifier')
        self.i += 1
        return c.s[2:] if c.s.startswith('r#') else c.s

    def split_shr(self):
        """The current token starts with '>' but is '>>', '>=', '>>=': split off one '>' (closing a generic list)."""
        c = self.cur
        rest = c.s[1:]
        self.t[self.i] = Tok('punct', '>', c.line)
        self.t.insert(self.i + 1, Tok('punct', rest, c.line))

    def close_angle(self):
        if self.cur.k == 'punct' and self.cur.s in ('>>', '>=', '>>='):
            self.split_shr()
        self.expect('>')

    def skip_balanced(self, open_s, close_s):
        """cur is just AFTER an opening delimiter; skip to after the matching closing one.  Returns the token range."""
        depth, start = 1, self.i
        while depth:
            c = self.cur
            if c.k == 'eof':
                self.err('unbalanced %s' % open_s)
            if c.k == 'punct':
                if c.s == open_s:
                    depth += 1
                elif c.s == close_s:
                    depth -= 1
            self.i += 1
        return start, self.i - 1

    def token_tree(self):
        """A delimited token tree starting at cur; returns (open delimiter, inner tokens)."""
        o = self.cur.s
        close = {'(': ')', '[': ']', '{': '}'}.get(o)
        if close is None or self.cur.k != 'punct':
            self.err('expected a delimited token tree')
        self.i += 1
        depth, start = 1, self.i
        while True:
            c = self.cur
            if c.k == 'eof':
                self.err('unbalanced token tree')
            if c.k == 'punct':
                if c.s in '([{' and len(c.s) == 1:
                    depth += 1
                elif c.s in ')]}' and len(c.s) == 1:
                    depth -= 1
                    if depth == 0:
                        break
            self.i += 1
        inner = self.t[start:self.i]
        self.i += 1
        return o, inner

    # ------------------------------------------------------------------ attributes, visibility
    def attributes(self):
        attrs = []
        while self.at('#'):
            self.i += 1
            self.eat('!')
            _, inner = self.token_tree()
            attrs.append(' '.join(x.s for x in inner))
        return attrs

    def visibility(self):
        if self.at_ident('pub'):
            self.i += 1
            if self.at('(') and self.peek().s in ('crate', 'super', 'self', 'in'):
                self.token_tree()

    # ------------------------------------------------------------------ items
    def parse_file(self):
        items = []
        while self.cur.k != 'eof':
            it = self.item()
            if it is not None:
                items.append(it)
        return items

    def item_end_recover(self, start):
        """Skip a whole item starting at token index `start` without understanding it: up to ';' or a '{...}' group."""
        self.i = start
        depth = 0
        while self.cur.k != 'eof':
            c = self.cur
            if c.k == 'punct':
                if c.s in ('(', '['):
                    depth += 1
                elif c.s in (')', ']'):
                    depth -= 1
                elif c.s == '{':
                    self.i += 1
                    self.skip_balanced('{', '}')
                    if depth <= 0:
                        self.eat(';')
                        return
                    continue
                elif c.s == ';' and depth <= 0:
                    self.i += 1
                    return
                elif c.s == '}' and depth <= 0:
                    return
            self.i += 1

    def item(self):
        start = self.i
        try:
            return self.item_inner()
        except ParseError as e:
            self.item_end_recover(start)
            if self.i == start:
                self.i += 1
            return ('unparsed', str(e))

    def item_inner(self):
        attrs = self.attributes()
        if self.cur.k == 'eof' or self.at('}'):
            return None
        self.visibility()
        c = self.cur
        if c.k != 'ident':
            if self.eat(';'):
                return None
            self.err('expected an item')
        kw = c.s
        while kw in ('unsafe', 'async', 'default') or (kw == 'extern' and self.peek().k == 'str') or \
                (kw == 'const' and self.peek().s in ('fn', 'unsafe')):
            self.i += 1
            if kw == 'extern':
                self.i += 1
            kw = self.cur.s
        if kw == 'use':
            self.i += 1
            try:
                self.use_tree([])
            except ParseError:
                pass
            self.item_end_recover(self.i - 1 if self.t[self.i - 1].s == ';' else self.i)
            return None
        if kw == 'extern' and self.peek().s == 'crate':
            self.item_end_recover(self.i)
            return None
        if kw == 'fn':
            return self.fn_item(attrs)
        if kw in ('const', 'static'):
            self.i += 1
            self.eat('mut')
            self.eat('ref')  # lazy_static's `static ref`
            name = self.ident() if not self.at('_') else self.ident()
            ty = None
            if self.eat(':'):
                ty = self.type()
            init = None
            if self.eat('='):
                init = self.expr()
            self.expect(';')
            return (kw, name, ty, init, attrs, self)
        if kw == 'mod':
            self.i += 1
            name = self.ident()
            if self.eat(';'):
                return None
            self.expect('{')
            items = []
            while not self.at('}'):
                if self.cur.k == 'eof':
                    self.err('unterminated mod')
                it = self.item()
                if it is not None:
                    items.append(it)
            self.expect('}')
            return ('mod', name, items, attrs)
        if kw == 'struct' or kw == 'union':
            return self.struct_item(attrs)
        if kw == 'enum':
            return self.enum_item(attrs)
        if kw == 'impl':
            return self.impl_item(attrs)
        if kw == 'trait':
            self.i += 1
            name = self.ident()
            # generics / bounds / where: skip to the body
            while not self.at('{'):
                if self.cur.k == 'eof':
                    self.err('trait without a body')
                if self.at('<'):
                    self.generics()
                else:
                    self.i += 1
            self.expect('{')
            items = []
            while not self.at('}'):
                it = self.item()
                if it is not None:
                    items.append(it)
            self.expect('}')
            return ('trait', name, items, attrs)
        if kw == 'type':
            self.item_end_recover(self.i)
            return None
        if kw == 'macro_rules' and self.peek().s == '!':
            self.i += 2
            name = self.ident()
            _, inner = self.token_tree()
            self.eat(';')
            return ('macro_rules', name, inner)
        if self.peek().s == '!' or (self.peek().s == '::' and self.peek(3).s == '!'):
            # item-position macro invocation: lazy_static! { ... }, user macros
            path = [self.ident()]
            while self.eat('::'):
                path.append(self.ident())
            self.expect('!')
            delim, inner = self.token_tree()
            if delim != '{':
                self.eat(';')
            return ('macro_item', path[-1], inner, attrs)
        self.err('unsupported item')

    def use_tree(self, prefix):
        """Record `use` aliases of this file: last segment (or `as` name) -> full path."""
        segs = list(prefix)
        self.eat('::')
        while True:
            if self.at('{'):
                self.i += 1
                while not self.at('}'):
                    self.use_tree(segs)
                    if not self.eat(','):
                        break
                self.expect('}')
                return
            if self.at('*'):
                self.i += 1
                return
            segs.append(self.ident())
            if self.at('::'):
                self.i += 1
                continue
            break
        alias = segs[-1]
        if self.at_ident('as'):
            self.i += 1
            alias = self.ident()
        if alias == 'self':
            segs = segs[:-1]
            alias = segs[-1]
        self.uses[alias] = segs

    def generics(self):
        """cur is '<' (or not: returns []).  Returns [(kind, name)] with kind in 'const', 'type', 'lifetime'."""
        out = []
        if not self.at('<'):
            return out
        self.i += 1
        while True:
            if self.cur.k == 'punct' and self.cur.s in ('>', '>>', '>=', '>>='):
                self.close_angle()
                break
            self.attributes()
            if self.cur.k == 'lifetime':
                out.append(('lifetime', self.cur.s))
                self.i += 1
            elif self.at_ident('const'):
                self.i += 1
                name = self.ident()
                self.expect(':')
                self.type()
                out.append(('const', name))
            else:
                out.append(('type', self.ident()))
            # bounds and defaults: skip to the next ',' or the closing '>' at depth 0
            depth = 0
            while True:
                c = self.cur
                if c.k == 'eof':
                    self.err('unterminated generics')
                if c.k == 'punct':
                    if c.s in ('<', '(', '['):
                        depth += 1
                    elif c.s in (')', ']'):
                        depth -= 1
                    elif c.s in ('>', '>>', '>=', '>>='):
                        if depth == 0:
                            break
                        if c.s != '>':
                            self.split_shr()
                        depth -= 1
                    elif c.s == ',' and depth == 0:
                        break
                self.i += 1
            self.eat(',')
        return out

    def where_clause(self):
        if self.at_ident('where'):
            depth = 0
            while True:
                c = self.cur
                if c.k == 'eof':
                    self.err('unterminated where clause')
                if c.k == 'punct':
                    if c.s in ('(', '[', '<'):
                        depth += 1
                    elif c.s in (')', ']', '>'):
                        depth -= 1
                    elif c.s == '>>':
                        depth -= 2
                    elif c.s in ('{', ';') and depth <= 0:
                        return
                self.i += 1

    def fn_item(self, attrs):
        self.expect('fn')
        name = self.ident()
        gen = self.generics()
        self.expect('(')
        params, self_kind = [], None
        while not self.at(')'):
            self.attributes()
            # self forms: self, mut self, &self, &mut self, &'a self, self: T
            j = self.i
            amp = False
            if self.at('&'):
                amp = True
                j += 1
                if self.t[j].k == 'lifetime':
                    j += 1
            mut = False
            if self.t[j].s == 'mut' and self.t[j].k == 'ident':
                mut = True
                j += 1
            if self.t[j].k == 'ident' and self.t[j].s == 'self' and self.t[j + 1].s in (',', ')', ':'):
                self.i = j + 1
                if self.eat(':'):
                    self.type()
                self_kind = ('ref_mut' if mut else 'ref') if amp else 'value'
            else:
                pat = self.pattern()
                self.expect(':')
                ty = self.type()
                params.append((pat, ty))
            if not self.eat(','):
                break
        self.expect(')')
        ret = None
        if self.eat('->'):
            ret = self.type()
        self.where_clause()
        body = None
        if self.eat(';'):
            pass
        else:
            self.expect('{')
            s, e = self.skip_balanced('{', '}')
            # the body's tokens up to and including the closing brace, parsed on first call.  A COPY, not an index range:
            # parsing a body can split a `>>` token in two (split_shr), which would shift every range recorded behind it
            body = self.t[s:e + 1]
        return ('fn', name, gen, params, self_kind, ret, body, attrs, self)

    def parse_body(self, rng):
        """Parse a function body from its tokens (lazily, on first call)."""
        if isinstance(rng, list):
            sub = Parser(rng + [Tok('eof', '', rng[-1].line)], self.fname)
            sub.uses = self.uses
            return sub.block_body()
        save = self.i
        self.i = rng[0]
        try:
            blk = self.block_body(end_index=rng[1])
        finally:
            self.i = save
        return blk

    def struct_item(self, attrs):
        self.i += 1
        name = self.ident()
        gen = self.generics()
        self.where_clause()
        if self.eat(';'):
            return ('struct', name, 'unit', [], attrs, gen)
        if self.at('('):
            self.i += 1
            fields = []
            while not self.at(')'):
                self.attributes()
                self.visibility()
                fields.append((str(len(fields)), self.type()))
                if not self.eat(','):
                    break
            self.expect(')')
            self.where_clause()
            self.eat(';')
            return ('struct', name, 'tuple', fields, attrs, gen)
        self.expect('{')
        fields = []
        while not self.at('}'):
            self.attributes()
            self.visibility()
            fname = self.ident()
            self.expect(':')
            fields.append((fname, self.type()))
            if not self.eat(','):
                break
        self.expect('}')
        return ('struct', name, 'named', fields, attrs, gen)

    def enum_item(self, attrs):
        self.i += 1
        name = self.ident()
        self.generics()
        self.where_clause()
        self.expect('{')
        variants, next_disc = [], 0
        while not self.at('}'):
            self.attributes()
            vname = self.ident()
            kind, fields, disc = 'unit', [], None
            if self.at('('):
                self.i += 1
                kind = 'tuple'
                while not self.at(')'):
                    self.attributes()
                    fields.append((str(len(fields)), self.type()))
                    if not self.eat(','):
                        break
                self.expect(')')
            elif self.at('{'):
                self.i += 1
                kind = 'named'
                while not self.at('}'):
                    self.attributes()
                    fn_ = self.ident()
                    self.expect(':')
                    fields.append((fn_, self.type()))
                    if not self.eat(','):
                        break
                self.expect('}')
            if self.eat('='):
                disc = self.expr()
            variants.append((vname, kind, fields, disc))
            if not self.eat(','):
                break
        self.expect('}')
        return ('enum', name, variants, attrs)

    def impl_item(self, attrs):
        self.expect('impl')
        self.generics()
        self.eat('!')
        t1 = self.type()
        trait = None
        if self.at_ident('for'):
            self.i += 1
            trait = t1
            t1 = self.type()
        self.where_clause()
        self.expect('{')
        items = []
        while not self.at('}'):
            if self.cur.k == 'eof':
                self.err('unterminated impl')
            it = self.item()
            if it is not None:
                items.append(it)
        self.expect('}')
        return ('impl', t1, trait, items, attrs)

    # ------------------------------------------------------------------ types
    def type(self):
        c = self.cur
        if c.k == 'punct':
            if c.s in ('&', '&&'):
                self.i += 1
                if self.cur.k == 'lifetime':
                    self.i += 1
                mut = self.eat('mut')
                inner = self.type()
                t = ('tref', mut, inner)
                return ('tref', False, t) if c.s == '&&' else t
            if c.s == '*':
                self.i += 1
                if not self.eat('const'):
                    self.expect('mut')
                return ('tptr', self.type())
            if c.s == '[':
                self.i += 1
                elem = self.type()
                if self.eat(';'):
                    n = self.expr()
                    self.expect(']')
                    return ('tarray', elem, n)
                self.expect(']')
                return ('tslice', elem)
            if c.s == '(':
                self.i += 1
                elems = []
                while not self.at(')'):
                    elems.append(self.type())
                    if not self.eat(','):
                        break
                self.expect(')')
                if len(elems) == 1 and self.t[self.i - 2].s != ',':
                    return elems[0]
                return ('ttuple', elems)
            if c.s == '!':
                self.i += 1
                return ('tnever',)
            if c.s == '<':  # <T as Trait>::Name
                self.i += 1
                t = self.type()
                if self.at_ident('as'):
                    self.i += 1
                    self.type()
                self.close_angle()
                segs = []
                while self.eat('::'):
                    segs.append(self.ident())
                return ('tqual', t, segs)
            if c.s == '_':
                self.i += 1
                return ('tinfer',)
        if c.k == 'ident':
            if c.s == '_':
                self.i += 1
                return ('tinfer',)
            if c.s in ('impl', 'dyn'):
                self.i += 1
                self.bounds()
                return ('topaque',)
            if c.s in ('fn', 'unsafe', 'extern') and (c.s != 'fn' or self.peek().s == '('):
                while self.cur.s in ('unsafe', 'extern') or self.cur.k == 'str':
                    self.i += 1
                self.expect('fn')
                self.token_tree()
                if self.eat('->'):
                    self.type()
                return ('tfn',)
            if c.s == 'for' and self.peek().s == '<':
                self.i += 1
                self.generics()
                return self.type()
            return self.type_path()
        if c.k == 'lifetime':
            self.i += 1
            return ('tlifetime',)
        self.err('expected a type')

    def bounds(self):
        while True:
            if self.cur.k == 'lifetime':
                self.i += 1
            else:
                self.eat('?')
                if self.at('('):
                    self.token_tree()
                else:
                    self.type_path()
            if not self.eat('+'):
                break

    def type_path(self):
        segs, gargs = [], []
        self.eat('::')
        while True:
            segs.append(self.ident())
            if self.at('<') or (self.at('::') and self.peek().s == '<'):
                self.eat('::')
                gargs = self.generic_args()
            elif self.at('(') and segs[-1] in ('Fn', 'FnMut', 'FnOnce'):
                self.token_tree()
                if self.eat('->'):
                    self.type()
            if self.at('::') and self.peek().k == 'ident':
                self.i += 1
                continue
            break
        return ('tpath', segs, gargs)

    def generic_args(self):
        """cur is '<'.  Each argument is a type, a lifetime, a const expression (literal / block / -literal) or
        `Name = Type`.  Returns the list of ('gtype', type) / ('gconst', expr)."""
        self.expect('<')
        out = []
        while True:
            if self.cur.k == 'punct' and self.cur.s in ('>', '>>', '>=', '>>='):
                break
            c = self.cur
            if c.k == 'lifetime':
                self.i += 1
            elif c.k in ('int', 'float', 'str', 'char') or c.s in ('true', 'false', '-') and c.k in ('ident', 'punct') and (c.s != '-' or self.peek().k in ('int', 'float')):
                out.append(('gconst', self.unary()))
            elif c.k == 'punct' and c.s == '{':
                out.append(('gconst', self.block_expr()))
            else:
                t = self.type()
                if self.eat('='):  # associated type binding
                    self.type()
                else:
                    out.append(('gtype', t))
            if not self.eat(','):
                break
        self.close_angle()
        return out

    # ------------------------------------------------------------------ patterns
    def pattern(self):
        self.eat('|')
        p = self.pattern_one()
        if self.at('|'):
            alts = [p]
            while self.eat('|'):
                alts.append(self.pattern_one())
            return ('por', alts)
        return p

    def pattern_literal(self):
        neg = self.eat('-')
        c = self.cur
        if c.k == 'int' or c.k == 'float':
            e = self.literal()
            return ('unary', '-', e) if neg else e
        if c.k in ('str', 'char'):
            return self.literal()
        if c.k == 'ident' and c.s in ('true', 'false'):
            self.i += 1
            return ('bool', c.s == 'true')
        self.err('expected a literal pattern')

    def pattern_one(self):
        c = self.cur
        if c.k == 'punct':
            if c.s == '_':
                self.i += 1
                return ('pwild',)
            if c.s in ('&', '&&'):
                self.i += 1
                self.eat('mut')
                inner = self.pattern_one()
                p = ('pref', inner)
                return ('pref', p) if c.s == '&&' else p
            if c.s == '(':
                self.i += 1
                elems = []
                while not self.at(')'):
                    elems.append(self.pattern())
                    if not self.eat(','):
                        break
                self.expect(')')
                if len(elems) == 1 and self.t[self.i - 2].s != ',':
                    return elems[0]
                return ('ptuple', elems)
            if c.s == '[':
                self.i += 1
                elems = []
                while not self.at(']'):
                    elems.append(self.pattern())
                    if not self.eat(','):
                        break
                self.expect(']')
                return ('pslice', elems)
            if c.s == '..':
                self.i += 1
                if self.cur.k in ('int', 'float', 'char'):  # ..=hi
                    return ('prange', None, self.pattern_literal(), True)
                return ('prest',)
            if c.s == '..=':
                self.i += 1
                return ('prange', None, self.pattern_literal(), True)
            if c.s == '-':
                lo = self.pattern_literal()
                return self.pattern_range_tail(lo)
        if c.k in ('int', 'float', 'char', 'str'):
            lo = self.pattern_literal()
            return self.pattern_range_tail(lo)
        if c.k == 'ident':
            if c.s == '_':
                self.i += 1
                return ('pwild',)
            if c.s in ('true', 'false'):
                return ('plit', self.pattern_literal())
            by_ref = mut = False
            if c.s == 'ref':
                self.i += 1
                by_ref = True
            if self.at_ident('mut'):
                self.i += 1
                mut = True
            if by_ref or mut:
                name = self.ident()
                sub = self.pattern_one() if self.eat('@') else None
                return ('pbind', name, sub)
            if c.s == 'box':
                self.i += 1
                return self.pattern_one()
            # path or binding
            path = self.expr_path_segments()
            if self.at('('):
                self.i += 1
                elems = []
                while not self.at(')'):
                    elems.append(self.pattern())
                    if not self.eat(','):
                        break
                self.expect(')')
                return ('ptstruct', path, elems)
            if self.at('{'):
                self.i += 1
                fields, rest = [], False
                while not self.at('}'):
                    self.attributes()
                    if self.eat('..'):
                        rest = True
                        break
                    self.eat('ref')
                    self.eat('mut')
                    fname = self.cur.s if self.cur.k == 'int' else self.ident()
                    if self.cur.k == 'int':
                        self.i += 1
                    if self.eat(':'):
                        fields.append((fname, self.pattern()))
                    else:
                        fields.append((fname, ('pbind', fname, None)))
                    if not self.eat(','):
                        break
                self.expect('}')
                return ('pstruct', path, fields, rest)
            if len(path) == 1:
                if self.eat('@'):
                    return ('pbind', path[0], self.pattern_one())
                if self.at('..=') or self.at('..'):
                    return self.pattern_range_tail(('path', path, None))
                return ('pident', path[0])  # binding or constant: decided at run time by name lookup
            p = ('ppath', path)
            if self.at('..=') or (self.at('..') and self.peek().k in ('int', 'float', 'char', 'ident')):
                return self.pattern_range_tail(('path', path, None))
            return p
        self.err('expected a pattern')

    def pattern_range_tail(self, lo):
        if self.at('..=') or self.at('...'):
            self.i += 1
            hi = self.pattern_range_end()
            return ('prange', lo, hi, True)
        if self.at('..'):
            self.i += 1
            if self.cur.k in ('int', 'float', 'char') or self.at('-') or self.cur.k == 'ident' and self.cur.s not in ('if',) and self.cur.s[0].isupper():
                return ('prange', lo, self.pattern_range_end(), False)
            return ('prange', lo, None, False)
        return ('plit', lo)

    def pattern_range_end(self):
        if self.cur.k == 'ident':
            return ('path', self.expr_path_segments(), None)
        return self.pattern_literal()

    # ------------------------------------------------------------------ expressions
    def literal(self):
        c = self.cur
        self.i += 1
        if c.k == 'int':
            s = c.s.replace('_', '')
            m = INT_SUFFIX.search(s)
            suffix = None
            if m and not (s.startswith('0x') and m.start() <= 2):
                suffix = m.group(1)
                s = s[:m.start()]
            v = int(s[2:], 16) if s.startswith('0x') else int(s[2:], 2) if s.startswith('0b') else int(s[2:], 8) if s.startswith('0o') else int(s)
            return ('int', v, suffix)
        if c.k == 'float':
            s = c.s.replace('_', '')
            suffix = None
            if s.endswith('f32') or s.endswith('f64'):
                suffix, s = s[-3:], s[:-3]
            return ('float', s, suffix)
        if c.k == 'str':
            return ('str', c.s)
        if c.k == 'char':
            body = c.s[c.s.index("'") + 1:-1]
            if body.startswith('\\'):
                esc = {'n': '\n', 't': '\t', 'r': '\r', '0': '\0', '\\': '\\', "'": "'", '"': '"'}
                if body[1] == 'x':
                    ch = chr(int(body[2:], 16))
                elif body[1] == 'u':
                    ch = chr(int(body[3:-1], 16))
                else:
                    ch = esc[body[1]]
            else:
                ch = body
            return ('char', ord(ch), c.s.startswith('b'))
        self.err('expected a literal')

    def expr(self, no_struct=False):
        return self.assign_expr(no_struct)

    def assign_expr(self, no_struct):
        c = self.cur
        if c.k == 'ident':
            if c.s == 'return':
                self.i += 1
                e = None
                if not self.expr_ends():
                    e = self.expr(no_struct)
                return ('return', e)
            if c.s == 'break':
                self.i += 1
                label = None
                if self.cur.k == 'lifetime':
                    label = self.cur.s
                    self.i += 1
                e = None
                if not self.expr_ends():
                    e = self.expr(no_struct)
                return ('break', label, e)
            if c.s == 'continue':
                self.i += 1
                label = None
                if self.cur.k == 'lifetime':
                    label = self.cur.s
                    self.i += 1
                return ('continue', label)
        if (c.k == 'punct' and c.s in ('|', '||')) or (c.k == 'ident' and c.s == 'move' and self.peek().s in ('|', '||')):
            return self.closure(no_struct)
        lhs = self.range_expr(no_struct)
        c = self.cur
        if c.k == 'punct' and c.s in ASSIGN_OPS:
            self.i += 1
            rhs = self.assign_expr(no_struct)
            if c.s == '=':
                return ('assign', lhs, rhs)
            return ('opassign', c.s[:-1], lhs, rhs)
        return lhs

    def expr_ends(self):
        c = self.cur
        return c.k == 'eof' or (c.k == 'punct' and c.s in (';', '}', ')', ']', ',', '=>'))

    def closure(self, no_struct):
        self.eat('move')
        params = []
        if self.eat('||'):
            pass
        else:
            self.expect('|')
            while not self.at('|'):
                p = self.pattern_one()
                ty = self.type() if self.eat(':') else None
                params.append((p, ty))
                if not self.eat(','):
                    break
            self.expect('|')
        if self.eat('->'):
            self.type()
            body = self.block_expr()
        else:
            body = self.expr(no_struct)
        return ('closure', params, body)

    def range_expr(self, no_struct):
        c = self.cur
        if c.k == 'punct' and c.s in ('..', '..='):
            self.i += 1
            hi = None
            if not self.expr_ends() and not (no_struct and self.at('{')):
                hi = self.binary(no_struct, 0)
            return ('range', None, hi, c.s == '..=')
        lo = self.binary(no_struct, 0)
        c = self.cur
        if c.k == 'punct' and c.s in ('..', '..='):
            self.i += 1
            hi = None
            if not self.expr_ends() and not (self.at('{') and no_struct):
                hi = self.binary(no_struct, 0)
            return ('range', lo, hi, c.s == '..=')
        return lo

    def binary(self, no_struct, min_prec):
        lhs = self.cast_expr(no_struct)
        while True:
            c = self.cur
            if c.k != 'punct':
                break
            prec = BIN_PREC.get(c.s)
            if prec is None or prec <= min_prec:
                break
            self.i += 1
            rhs = self.binary(no_struct, prec)  # left-associative: the right operand takes tighter operators only
            if c.s == '&&':
                lhs = ('and', lhs, rhs)
            elif c.s == '||':
                lhs = ('or', lhs, rhs)
            else:
                lhs = ('binary', c.s, lhs, rhs)
        return lhs

    def cast_expr(self, no_struct):
        e = self.unary(no_struct)
        while self.at_ident('as'):
            self.i += 1
            e = ('cast', e, self.type())
        return e

    def unary(self, no_struct=False):
        c = self.cur
        if c.k == 'punct':
            if c.s in ('-', '!'):
                self.i += 1
                return ('unary', c.s, self.unary(no_struct))
            if c.s == '*':
                self.i += 1
                return ('deref', self.unary(no_struct))
            if c.s in ('&', '&&'):
                self.i += 1
                mut = self.eat('mut')
                e = ('ref', mut, self.unary(no_struct))
                return ('ref', False, e) if c.s == '&&' else e
        return self.postfix(no_struct)

    def postfix(self, no_struct, start=None):
        e = self.primary(no_struct) if start is None else start
        while True:
            c = self.cur
            if c.k != 'punct':
                break
            if c.s == '.':
                nx = self.peek()
                if nx.k == 'int':
                    self.i += 2
                    e = ('field', e, nx.s)
                    continue
                if nx.k == 'float' and re.match(r'^\d+\.\d+$', nx.s):  # x.0.1
                    self.i += 2
                    a, b = nx.s.split('.')
                    e = ('field', ('field', e, a), b)
                    continue
                if nx.k == 'ident':
                    self.i += 1
                    if nx.s == 'await':
                        self.i += 1
                        continue
                    name = self.ident()
                    gargs = []
                    if self.at('::'):
                        self.i += 1
                        gargs = self.generic_args()
                    if self.at('('):
                        e = ('mcall', e, name, gargs, self.call_args())
                    else:
                        e = ('field', e, name)
                    continue
                break
            if c.s == '(':
                e = ('call', e, self.call_args())
                continue
            if c.s == '[':
                self.i += 1
                idx = self.expr()
                self.expect(']')
                e = ('index', e, idx)
                continue
            if c.s == '?':
                self.i += 1
                e = ('try', e)
                continue
            break
        return e

    def call_args(self):
        self.expect('(')
        args = []
        while not self.at(')'):
            args.append(self.expr())
            if not self.eat(','):
                break
        self.expect(')')
        return args

    def expr_path_segments(self):
        segs = []
        self.eat('::')
        while True:
            segs.append(self.ident())
            if self.at('::') and self.peek().k == 'ident':
                self.i += 1
                continue
            break
        return segs

    def primary(self, no_struct):
        c = self.cur
        if c.k in ('int', 'float', 'str', 'char'):
            return self.literal()
        if c.k == 'lifetime' and self.peek().s == ':':  # labelled loop
            label = c.s
            self.i += 2
            e = self.primary(no_struct)
            return ('labelled', label, e)
        if c.k == 'punct':
            if c.s == '(':
                self.i += 1
                elems = []
                trailing = False
                while not self.at(')'):
                    elems.append(self.expr())
                    trailing = False
                    if not self.eat(','):
                        break
                    trailing = True
                self.expect(')')
                if len(elems) == 1 and not trailing:
                    return ('paren', elems[0])
                return ('tuple', elems)
            if c.s == '[':
                self.i += 1
                if self.at(']'):
                    self.i += 1
                    return ('array', [])
                first = self.expr()
                if self.eat(';'):
                    n = self.expr()
                    self.expect(']')
                    return ('repeat', first, n)
                elems = [first]
                while self.eat(','):
                    if self.at(']'):
                        break
                    elems.append(self.expr())
                self.expect(']')
                return ('array', elems)
            if c.s == '{':
                return self.block_expr()
            if c.s == '<':  # qualified path <T as Trait>::f / <T>::f
                self.i += 1
                t = self.type()
                if self.at_ident('as'):
                    self.i += 1
                    self.type()
                self.close_angle()
                segs = []
                while self.eat('::'):
                    segs.append(self.ident())
                return ('qpath', t, segs)
            if c.s == '::':
                self.i += 1
                return self.primary(no_struct)
            if c.s == '#':
                self.attributes()
                return self.primary(no_struct)
        if c.k == 'ident':
            s = c.s
            if s in ('true', 'false'):
                self.i += 1
                return ('bool', s == 'true')
            if s == 'if':
                return self.if_expr()
            if s == 'match':
                self.i += 1
                scrut = self.expr(no_struct=True)
                self.expect('{')
                arms = []
                while not self.at('}'):
                    self.attributes()
                    pat = self.pattern()
                    guard = None
                    if self.at_ident('if'):
                        self.i += 1
                        guard = self.expr()
                    self.expect('=>')
                    # an arm whose body is a block ends at the block's brace (`{ .. } (3, 0) => ..` is two arms, not a call)
                    body = self.block_expr() if self.at('{') else self.expr()
                    arms.append((pat, guard, body))
                    if not self.eat(','):
                        if self.at('}'):
                            break
                        # block-bodied arm without a comma
                        if body[0] not in ('block', 'if', 'iflet', 'match', 'loop', 'while', 'whilelet', 'for', 'unsafe'):
                            self.err("expected ',' after match arm")
                self.expect('}')
                return ('match', scrut, arms)
            if s == 'loop':
                self.i += 1
                return ('loop', self.block_expr())
            if s == 'while':
                self.i += 1
                if self.at_ident('let'):
                    self.i += 1
                    pat = self.pattern()
                    self.expect('=')
                    e = self.expr(no_struct=True)
                    return ('whilelet', pat, e, self.block_expr())
                cond = self.expr(no_struct=True)
                return ('while', cond, self.block_expr())
            if s == 'for':
                self.i += 1
                pat = self.pattern()
                if not self.at_ident('in'):
                    self.err("expected 'in'")
                self.i += 1
                it = self.expr(no_struct=True)
                return ('for', pat, it, self.block_expr())
            if s == 'unsafe' and self.peek().s == '{':
                self.i += 1
                return self.block_expr()
            if s == 'let':  # let in condition position (if let chains) is handled by if_expr
                self.err("unexpected 'let'")
            # path expression (with optional turbofish), struct literal, macro call
            segs, gargs = [], None
            while True:
                segs.append(self.ident())
                if self.at('::'):
                    if self.peek().s == '<':
                        self.i += 1
                        gargs = self.generic_args()
                        if self.at('::') and self.peek().k == 'ident':
                            self.i += 1
                            continue
                        break
                    if self.peek().k == 'ident':
                        self.i += 1
                        continue
                break
            if self.at('!') and self.peek().s in ('(', '[', '{') and not (self.peek().s == '=' ):
                self.i += 1
                delim, inner = self.token_tree()
                return ('macro', segs[-1], inner, delim)
            if self.at('{') and not no_struct and self.looks_like_struct_literal(segs):
                self.i += 1
                fields, base = [], None
                while not self.at('}'):
                    self.attributes()
                    if self.eat('..'):
                        base = self.expr()
                        break
                    fname = self.cur.s if self.cur.k == 'int' else self.ident()
                    if self.cur.k == 'int':
                        self.i += 1
                    if self.eat(':'):
                        fields.append((fname, self.expr()))
                    else:
                        fields.append((fname, ('path', [fname], None)))
                    if not self.eat(','):
                        break
                self.expect('}')
                return ('struct', segs, fields, base)
            return ('path', segs, gargs)
        self.err('expected an expression')

    def looks_like_struct_literal(self, segs):
        # `Name {` followed by `ident :`, `ident ,`, `ident }`, `}` or `..`
        a, b = self.peek(1), self.peek(2)
        if not segs[-1][0].isupper() and segs[-1] != 'Self':
            return False
        if a.s == '}' or a.s == '..':
            return True
        if a.k in ('ident', 'int') and b.s in (':', ',', '}') and not (b.s == ':' and self.peek(3).s == ':'):
            return True
        return False

    def if_expr(self):
        self.expect('if')
        if self.at_ident('let'):
            self.i += 1
            pat = self.pattern()
            self.expect('=')
            e = self.expr(no_struct=True)
            then = self.block_expr()
            els = self.else_part()
            return ('iflet', pat, e, then, els)
        cond = self.expr(no_struct=True)
        then = self.block_expr()
        return ('if', cond, then, self.else_part())

    def else_part(self):
        if self.at_ident('else'):
            self.i += 1
            if self.at_ident('if'):
                return self.if_expr()
            return self.block_expr()
        return None

    def block_expr(self):
        self.expect('{')
        return self.block_body()

    def block_body(self, end_index=None):
        """Statements up to the closing '}' (consumed).  Returns ('block', stmts, tail_expr_or_None)."""
        stmts, tail = [], None
        while True:
            if self.at('}') and (end_index is None or self.i >= end_index):
                self.i += 1
                break
            if self.cur.k == 'eof':
                self.err('unterminated block')
            if self.eat(';'):
                continue
            attrs = self.attributes() if self.at('#') else []
            c = self.cur
            if c.k == 'ident' and c.s == 'let':
                self.i += 1
                pat = self.pattern()
                ty = self.type() if self.eat(':') else None
                init = els = None
                if self.eat('='):
                    init = self.expr()
                    if self.at_ident('else'):
                        self.i += 1
                        els = self.block_expr()
                self.expect(';')
                stmts.append(('let', pat, ty, init, els))
                continue
            if c.k == 'ident' and (c.s in ('fn', 'struct', 'enum', 'impl', 'trait', 'use', 'mod', 'type', 'pub', 'extern', 'static') or
                                   (c.s == 'const' and self.peek().k == 'ident' and self.peek().s != 'fn' and self.peek(2).s in (':', '=')) or
                                   (c.s == 'const' and self.peek().s == 'fn') or
                                   (c.s == 'macro_rules' and self.peek().s == '!')):
                it = self.item_inner()
                if it is not None:
                    stmts.append(('item', it))
                continue
            # expression statement; a block-like expression at statement position is a statement by itself
            if (c.k == 'ident' and c.s in ('if', 'match', 'loop', 'while', 'for', 'unsafe')) or (c.k == 'punct' and c.s == '{') or \
                    (c.k == 'lifetime' and self.peek().s == ':'):
                e = self.primary(False)
                if self.at('.') or self.at('?'):  # `match x { .. }?;` / `if c { a } else { b }.f();`: the statement goes on
                    e = self.postfix(False, start=e)
            else:
                e = self.expr()
            if self.eat(';'):
                stmts.append(('expr', e))
                continue
            if self.at('}') and (end_index is None or self.i >= end_index):
                tail = e
                continue
            if e[0] in ('if', 'iflet', 'match', 'loop', 'while', 'whilelet', 'for', 'block', 'labelled') or (e[0] == 'macro' and e[3] == '{'):
                stmts.append(('expr', e))
                continue
            self.err("expected ';' or '}' after expression")
        return ('block', stmts, tail)


def parse_source(src, fname='<src>'):
    p = Parser(lex(src), fname)
    return p.parse_file()


def parse_tokens_as_expr(toks, fname='<macro>'):
    p = Parser(list(toks) + [Tok('eof', '', toks[-1].line if toks else 0)], fname)
    e = p.expr()
    if p.cur.k != 'eof':
        p.err('trailing tokens after expression')
    return e


def parse_tokens_as_items(toks, fname='<macro>'):
    p = Parser(list(toks) + [Tok('eof', '', toks[-1].line if toks else 0)], fname)
    return p.parse_file()


def split_commas(toks):
    """Split a token list at top-level commas."""
    out, cur, depth = [], [], 0
    for t in toks:
        if t.k == 'punct':
            if t.s in ('(', '[', '{'):
                depth += 1
            elif t.s in (')', ']', '}'):
                depth -= 1
            elif t.s == ',' and depth == 0:
                out.append(cur)
                cur = []
                continue
        cur.append(t)
    if cur:
        out.append(cur)
    return out
