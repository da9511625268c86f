"""Minimal HOCON-subset reader with the pyhocon ConfigTree surface the model code uses.

The reference parses code/confs_sg/*.conf with pyhocon (idr_train.py:42) and the model only calls
get_int / get_float / get_bool / get_string / get_config / get_list and ``**conf.get_config(k)``
(implicit_differentiable_renderer.py:247-258).  pyhocon is not installed on the build or GPU image,
so this module parses the subset those files use: ``key = value``, ``key { ... }`` (with or without '='),
lists, numbers, booleans, bare / quoted strings and '#' or '//' comments.
"""
import re


class ConfigTree(dict):
    def _get(self, key, default=None, required=True):
        cur = self
        for part in key.split('.'):
            if not isinstance(cur, dict) or part not in cur:
                if default is not None or not required:
                    return default
                raise KeyError('No configuration setting found for key ' + key)
            cur = cur[part]
        return cur

    def get(self, key, default=None):
        return self._get(key, default, required=False)

    def get_int(self, key, default=None):
        return int(self._get(key, default))

    def get_float(self, key, default=None):
        return float(self._get(key, default))

    def get_bool(self, key, default=None):
        v = self._get(key, default)
        if isinstance(v, str):
            return v.lower() in ('true', 'yes', 'on', '1')
        return bool(v)

    def get_string(self, key, default=None):
        return str(self._get(key, default))

    def get_list(self, key, default=None):
        return list(self._get(key, default))

    def get_config(self, key, default=None):
        v = self._get(key, default)
        return v if isinstance(v, ConfigTree) else ConfigTree(v)


_TOKEN = re.compile(r'''\s*(?:(?P<brace>[{}\[\],=:])|"(?P<dq>[^"]*)"|'(?P<sq>[^']*)'|(?P<bare>[^\s{}\[\],=:#"']+))''')


def _strip_comment(line):
    """cut the line at the first '#' or '//' that is not inside a quoted string"""
    quote = None
    for i, ch in enumerate(line):
        if quote:
            if ch == quote:
                quote = None
        elif ch in '"\'':
            quote = ch
        elif ch == '#' or line.startswith('//', i):
            return line[:i]
    return line


def _tokens(text):
    out = []
    for line in text.splitlines():
        line = _strip_comment(line)
        pos = 0
        while pos < len(line):
            m = _TOKEN.match(line, pos)
            if not m:
                break
            pos = m.end()
            if m.group('brace'):
                out.append(('p', m.group('brace')))
            elif m.group('dq') is not None:
                out.append(('s', m.group('dq')))
            elif m.group('sq') is not None:
                out.append(('s', m.group('sq')))
            elif m.group('bare') is not None:
                out.append(('b', m.group('bare')))
        out.append(('p', '\n'))
    return out


def _scalar(tok):
    kind, v = tok
    if kind == 's':
        return v
    low = v.lower()
    if low in ('true', 'false'):
        return low == 'true'
    if low in ('null', 'none'):
        return None
    try:
        return int(v)
    except ValueError:
        pass
    try:
        return float(v)
    except ValueError:
        return v


class _Parser:
    def __init__(self, toks):
        self.t, self.i = toks, 0

    def peek(self):
        return self.t[self.i] if self.i < len(self.t) else None

    def skip_nl(self):
        while self.peek() in (('p', '\n'), ('p', ',')):
            self.i += 1

    def obj(self, top=False):
        tree = ConfigTree()
        while True:
            self.skip_nl()
            tok = self.peek()
            if tok is None:
                if top:
                    return tree
                raise ValueError('unterminated {')
            if tok == ('p', '}'):
                self.i += 1
                return tree
            key = tok[1]
            self.i += 1
            self.skip_nl_only()
            nxt = self.peek()
            if nxt in (('p', '='), ('p', ':')):
                self.i += 1
                self.skip_nl_only()
            val = self.value()
            cur = tree
            parts = key.split('.')
            for p in parts[:-1]:
                cur = cur.setdefault(p, ConfigTree())
            if isinstance(val, ConfigTree) and isinstance(cur.get(parts[-1]), ConfigTree):
                cur[parts[-1]].update(val)
            else:
                cur[parts[-1]] = val

    def skip_nl_only(self):
        while self.peek() == ('p', '\n'):
            self.i += 1

    def value(self):
        tok = self.peek()
        if tok == ('p', '{'):
            self.i += 1
            return self.obj()
        if tok == ('p', '['):
            self.i += 1
            items = []
            while True:
                self.skip_nl()
                if self.peek() == ('p', ']'):
                    self.i += 1
                    return items
                items.append(self.value())
        self.i += 1
        return _scalar(tok)


def parse_string(text):
    return _Parser(_tokens(text)).obj(top=True)


def parse_file(path):
    with open(path) as f:
        return parse_string(f.read())


def from_dict(d):
    """Nested dict -> ConfigTree (used by the synthetic workloads)."""
    t = ConfigTree()
    for k, v in d.items():
        t[k] = from_dict(v) if isinstance(v, dict) else v
    return t


class ConfigFactory:          # pyhocon-compatible entry points
    parse_file = staticmethod(parse_file)
    parse_string = staticmethod(parse_string)
    from_dict = staticmethod(from_dict)
