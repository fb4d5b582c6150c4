"""Minimal HOCON-subset reader for confs/*.conf (the reference uses pyhocon, exp_runner.py:33-39).

Supported: nested `key { ... }` / `key = { ... }`, `key = value` / `key : value`, lists `[a, b]`, optional commas,
bare or quoted strings, ints, floats (5e-4), True/False/true/false, `#` and `//` comments.  Dotted lookups
(`conf['model.sdf_network']`), get_int / get_float / get_bool / get_string with defaults, like pyhocon's ConfigTree.
"""
from __future__ import annotations

import re
from collections import OrderedDict

_TOKEN = re.compile(r"""
    (?P<ws>[ \t\r]+) | (?P<nl>\n) | (?P<comment>(\#|//)[^\n]*) |
    (?P<lbrace>\{) | (?P<rbrace>\}) | (?P<lbrack>\[) | (?P<rbrack>\]) | (?P<comma>,) | (?P<eq>[=:]) |
    (?P<qstr>"(?:[^"\\]|\\.)*") | (?P<bare>[^\s{}\[\],=:#"]+)
""", re.X)


class ConfigTree(OrderedDict):
    def _walk(self, key):
        node = self
        for part in key.split("."):
            if not isinstance(node, dict) or part not in node:
                raise KeyError(key)
            node = OrderedDict.__getitem__(node, part)
        return node

    def __getitem__(self, key):
        if isinstance(key, str) and "." in key and not OrderedDict.__contains__(self, key):
            return self._walk(key)
        return OrderedDict.__getitem__(self, key)

    def __setitem__(self, key, value):
        if isinstance(key, str) and "." in key:
            head, rest = key.split(".", 1)
            OrderedDict.__getitem__(self, head)[rest] = value
        else:
            OrderedDict.__setitem__(self, key, value)

    def __contains__(self, key):
        try:
            self[key]
            return True
        except KeyError:
            return False

    def get(self, key, default=None):
        try:
            return self[key]
        except KeyError:
            return default

    def get_int(self, key, default=None):
        v = self.get(key, default)
        return None if v is None else int(v)

    def get_float(self, key, default=None):
        v = self.get(key, default)
        return None if v is None else float(v)

    def get_bool(self, key, default=None):
        v = self.get(key, default)
        if isinstance(v, str):
            return v.lower() == "true"
        return None if v is None else bool(v)

    def get_string(self, key, default=None):
        v = self.get(key, default)
        return None if v is None else str(v)

    def get_list(self, key, default=None):
        return list(self.get(key, default))


def _scalar(tok: str):
    if tok in ("True", "true"):
        return True
    if tok in ("False", "false"):
        return False
    if tok in ("null", "None"):
        return None
    try:
        return int(tok)
    except ValueError:
        pass
    try:
        return float(tok)
    except ValueError:
        return tok


def _tokens(text):
    pos = 0
    while pos < len(text):
        m = _TOKEN.match(text, pos)
        if not m:
            raise ValueError(f"conf syntax error at offset {pos}: {text[pos:pos + 20]!r}")
        pos = m.end()
        kind = m.lastgroup
        if kind in ("ws", "comment"):
            continue
        yield kind, m.group(kind)
    yield "eof", ""


class _Parser:
    def __init__(self, text):
        self.toks = list(_tokens(text))
        self.i = 0

    def peek(self):
        return self.toks[self.i]

    def next(self):
        t = self.toks[self.i]
        self.i += 1
        return t

    def skip_sep(self):
        while self.peek()[0] in ("nl", "comma"):
            self.i += 1

    def parse_object(self, closing):
        tree = ConfigTree()
        while True:
            self.skip_sep()
            kind, val = self.peek()
            if kind == closing:
                self.next()
                return tree
            if kind not in ("bare", "qstr"):
                raise ValueError(f"conf: expected a key, got {val!r}")
            key = self.next()[1].strip('"')
            kind, val = self.peek()
            if kind == "lbrace":
                self.next()
                value = self.parse_object("rbrace")
            else:
                if kind != "eq":
                    raise ValueError(f"conf: expected '=' after {key!r}")
                self.next()
                value = self.parse_value()
            node = tree
            parts = key.split(".")
            for p in parts[:-1]:
                node = node.setdefault(p, ConfigTree())
            if isinstance(value, ConfigTree) and isinstance(node.get(parts[-1]), ConfigTree):
                node[parts[-1]].update(value)
            else:
                OrderedDict.__setitem__(node, parts[-1], value)

    def parse_value(self):
        while self.peek()[0] == "nl":
            self.next()
        kind, val = self.next()
        if kind == "lbrace":
            return self.parse_object("rbrace")
        if kind == "lbrack":
            out = []
            while True:
                self.skip_sep()
                if self.peek()[0] == "rbrack":
                    self.next()
                    return out
                out.append(self.parse_value())
        if kind == "qstr":
            return bytes(val[1:-1], "utf-8").decode("unicode_escape")
        if kind == "bare":
            # bare strings may continue to the end of the line (paths like ./exp/CASE/x)
            parts = [val]
            while self.peek()[0] == "bare":
                parts.append(self.next()[1])
            return _scalar(" ".join(parts)) if len(parts) > 1 else _scalar(val)
        raise ValueError(f"conf: unexpected token {val!r}")


def parse_string(text: str) -> ConfigTree:
    return _Parser(text).parse_object("eof")


def parse_file(path: str, case: str = None) -> ConfigTree:
    text = open(path).read()
    if case is not None:
        text = text.replace("CASE_NAME", case)      # exp_runner.py:35
    return parse_string(text)
