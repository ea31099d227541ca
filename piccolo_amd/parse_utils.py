"""Config interface of the reference (parse_utils.py:6-85): .ini -> flat namedtuple, `--override` value parsing.

Grammar kept verbatim because every consumer reads keys with getattr(cfg, key, default):
  * section names are dropped, keys are lower-cased by configparser;
  * a value that is all digits after removing at most one each of '.', '+', '-', 'e' is literal_eval'ed (int/float);
  * True/true/False/false -> bool (parse_value: only the capitalised forms); 'None' -> None;
  * a value containing ',' is a list, split on ', ' if present else on ','; numeric if the first item has a digit;
  * anything else stays a string.
"""
import configparser
import os
from ast import literal_eval
from collections import namedtuple


def _is_number(text):
    for ch in ".+-e":
        text = text.replace(ch, "", 1)
    return text.isdigit()


def _as_list(value, sep, strip):
    items = value.split(sep)
    numeric = any(ch.isdigit() for ch in items[0])
    if "" in items:
        items.remove("")            # only the first empty item, like list.remove
    if numeric:
        return [literal_eval(v) for v in items]
    return [v.strip() for v in items] if strip else list(items)


def parse_ini(config_path):
    reader = configparser.ConfigParser()
    reader.read(config_path)
    keys, values = [], {}
    for section in reader.sections():
        for key, raw in reader.items(section):
            keys.append(key)
            if _is_number(raw):
                values[key] = literal_eval(raw)
            elif raw in ("True", "true"):
                values[key] = True
            elif raw in ("False", "false"):
                values[key] = False
            elif raw == "None":
                values[key] = None
            elif "," in raw:
                values[key] = _as_list(raw, ", " if ", " in raw else ",", strip=False)
            else:
                values[key] = raw
    return namedtuple("Config", keys)(**values)


def parse_value(value):
    """Value grammar of `--override key=value` (main.py:24-45)."""
    if _is_number(value):
        return literal_eval(value)
    if value == "True":
        return True
    if value == "False":
        return False
    if value == "None":
        return None
    if "," in value:
        items = value.split(",")
        numeric = any(ch.isdigit() for ch in items[0])
        if "" in items:
            items.remove("")
        if numeric:
            return [literal_eval(v) for v in items]
        if '"' in items[0] and "'" in items[0]:
            return [literal_eval(v.strip()) for v in items]
        return [v.strip() for v in items]
    return value


def apply_override(cfg, override):
    """main.py:24-45: 'k=v' or 'k1=v1,k2=v2,...' (list values may be bracketed) -> new namedtuple."""
    parts = override.split("=")
    assert len(parts) > 0
    if len(parts) == 2:
        new = {parts[0]: parse_value(parts[1])}
    else:
        keys = [parts[0]] + [p.split(",")[-1] for p in parts[1:-1]]
        vals = [p.replace("," + k, "") for p, k in zip(parts[1:-1], keys[1:])] + [parts[-1]]
        vals = [v.replace("[", "").replace("]", "") for v in vals]
        new = {k: parse_value(v) for k, v in zip(keys, vals)}
    merged = cfg._asdict()
    merged.update(new)
    return namedtuple("Config", tuple(merged.keys()))(**merged)


def save_ini(config_path, log_path):
    reader = configparser.ConfigParser()
    reader.read(config_path)
    with open(os.path.join(log_path, "config.ini"), "w") as f:
        reader.write(f)
