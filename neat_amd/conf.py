"""Minimal ConfigTree with pyhocon's accessor API (get_int/get_float/get_bool/get_list/get_config/get_string)
plus a parser for the HOCON subset used by the reference's confs (code/confs/*.conf): nested `name { ... }` /
`name = { ... }` blocks, `key = value`, lists, `#` comments.  pyhocon itself is not available offline; a real
pyhocon ConfigTree can be passed to the model classes as well, the accessors are the same."""
import re

_MISSING = object()


class ConfTree(dict):
    def _get(self, key, default=_MISSING):
        node = self
        for part in key.split("."):
            if isinstance(node, dict) and part in node:
                node = node[part]
            elif default is _MISSING:
                raise KeyError(key)
            else:
                return default
        return node

    def get(self, key, default=None):
        return self._get(key, default)

    def get_int(self, key, default=_MISSING):
        return int(self._get(key, default))

    def get_float(self, key, default=_MISSING):
        return float(self._get(key, default))

    def get_bool(self, key, default=_MISSING):
        v = self._get(key, default)
        return v.lower() in ("true", "yes", "on") if isinstance(v, str) else bool(v)

    def get_string(self, key, default=_MISSING):
        return str(self._get(key, default))

    def get_list(self, key, default=_MISSING):
        return list(self._get(key, default))

    def get_config(self, key, default=_MISSING):
        v = self._get(key, default)
        return v if isinstance(v, ConfTree) else from_dict(v)


def from_dict(d):
    t = ConfTree()
    for k, v in d.items():
        t[k] = from_dict(v) if isinstance(v, dict) else v
    return t


_TOKEN = re.compile(r"""\s*(?:(\#[^\n]*|//[^\n]*)|([{}\[\],=:])|"([^"]*)"|([^\s{}\[\],=:#"]+))""")


def _scalar(tok):
    low = tok.lower()
    if low in ("true", "false"):
        return low == "true"
    if low in ("null", "none"):
        return None
    try:
        return int(tok)
    except ValueError:
        pass
    try:
        return float(tok)
    except ValueError:
        return tok


def parse_string(text):
    toks = []
    pos = 0
    while pos < len(text):
        m = _TOKEN.match(text, pos)
        if not m:
            if text[pos:].strip() == "":
                break
            raise ValueError(f"conf parse error at offset {pos}: {text[pos:pos + 30]!r}")
        pos = m.end()
        if m.group(1) is not None:
            continue
        if m.group(2) is not None:
            toks.append(("p", m.group(2)))
        elif m.group(3) is not None:
            toks.append(("s", m.group(3)))
        else:
            toks.append(("w", m.group(4)))
    it = iter(range(len(toks)))
    i = 0

    def parse_value():
        nonlocal i
        kind, tok = toks[i]
        if kind == "p" and tok == "{":
            i += 1
            return parse_object(closing=True)
        if kind == "p" and tok == "[":
            i += 1
            items = []
            while toks[i] != ("p", "]"):
                if toks[i] == ("p", ","):
                    i += 1
                    continue
                items.append(parse_value())
            i += 1
            return items
        i += 1
        return tok if kind == "s" else _scalar(tok)

    def parse_object(closing):
        nonlocal i
        obj = ConfTree()
        while i < len(toks):
            kind, tok = toks[i]
            if kind == "p" and tok == "}":
                if not closing:
                    raise ValueError("unbalanced '}'")
                i += 1
                return obj
            if kind == "p" and tok == ",":
                i += 1
                continue
            key = tok
            i += 1
            if toks[i][0] == "p" and toks[i][1] in "=:":
                i += 1
            val = parse_value()
            node = obj
            parts = key.split(".")
            for part in parts[:-1]:
                node = node.setdefault(part, ConfTree())
            if isinstance(val, ConfTree) and isinstance(node.get(parts[-1]), ConfTree):
                node[parts[-1]].update(val)
            else:
                node[parts[-1]] = val
        if closing:
            raise ValueError("missing '}'")
        return obj

    del it
    return parse_object(closing=False)


def parse_file(path):
    with open(path) as f:
        return parse_string(f.read())
