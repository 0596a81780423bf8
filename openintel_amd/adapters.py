"""The on-wire formats that feed the PostAnalyzer path (SURVEY.md section 8, row f rank 4).

Host mirror of the reference's response parsers -- JSON body in, `SocialPost`s out -- so that a
feed captured from the reference's sources can be replayed through the GPU analyzer (paths
relative to the openintel repo):

    reddit   parse_posts   src/adapters/sources/reddit/response.rs:49-98
    bluesky  parse_posts   src/adapters/sources/bluesky/response.rs:59-114

Only the parsing is mirrored: the HTTPS clients, rate limiting and credentials around it are out
of scope.  Same skip / truncate / fallback rules and the same failure
(`SourceFailure{name, "malformed response: ..."}`); the text after "malformed response:" is
Python's wording, not serde_json's.
"""
from __future__ import annotations

import datetime as _dt
import json
from typing import List, Optional

from .domain import MAX_POST_LEN, InvalidPostText, PostText, SocialPost, SourceFailure, SourceKind

_UNKNOWN = "[unknown]"
_U32_MAX = 0xFFFFFFFF
_I64_MIN, _I64_MAX = -(1 << 63), (1 << 63) - 1


class _Malformed(Exception):
    pass


def _no_constants(name):  # serde_json rejects NaN / Infinity literals
    raise _Malformed("invalid literal %s" % name)


def _load(body: str):
    try:
        return json.loads(body, parse_constant=_no_constants)
    except _Malformed:
        raise
    except (ValueError, RecursionError) as e:
        raise _Malformed(str(e))


def _obj(v, what: str) -> dict:
    if not isinstance(v, dict):
        raise _Malformed("%s: expected an object" % what)
    return v


def _opt_str(d: dict, key: str) -> Optional[str]:
    v = d.get(key)
    if v is None:
        return None
    if not isinstance(v, str):
        raise _Malformed("%s: expected a string" % key)
    return v


def _opt_i64(d: dict, key: str) -> Optional[int]:
    v = d.get(key)
    if v is None:
        return None
    if isinstance(v, bool) or not isinstance(v, int) or not (_I64_MIN <= v <= _I64_MAX):
        raise _Malformed("%s: expected an i64" % key)  # a float, even 1.0, is not an i64 to serde
    return v


def _opt_f64(d: dict, key: str) -> Optional[float]:
    v = d.get(key)
    if v is None:
        return None
    if isinstance(v, bool) or not isinstance(v, (int, float)):
        raise _Malformed("%s: expected a number" % key)
    return float(v)


def _f64_as_i64(x: float) -> int:
    """Rust `x as i64`: truncates toward zero, saturates, NaN -> 0."""
    if x != x:
        return 0
    if x >= 9.223372036854775807e18:
        return _I64_MAX
    if x <= -9.223372036854775808e18:
        return _I64_MIN
    return int(x)


def _timestamp(secs: int) -> Optional[_dt.datetime]:
    """Utc.timestamp_opt(secs, 0).single(): None outside chrono's range (about +-262000 years);
    Python's own range (years 1..9999) is narrower, outside it the fallback is taken as well."""
    try:
        return _dt.datetime.fromtimestamp(secs, tz=_dt.timezone.utc)
    except (OverflowError, OSError, ValueError):
        return None


def _parse_rfc3339(s: Optional[str]) -> Optional[_dt.datetime]:
    """DateTime::parse_from_rfc3339(..).ok() converted to UTC."""
    if s is None or len(s) < 20 or s[10] not in "Tt ":
        return None
    t = s[:10] + "T" + s[11:]
    if t[-1] in "zZ":
        t = t[:-1] + "+00:00"
    # fractional seconds: datetime.fromisoformat (3.10) wants exactly 3 or 6 digits
    if "." in t:
        head, rest = t.split(".", 1)
        i = 0
        while i < len(rest) and rest[i].isdigit():
            i += 1
        if i == 0:
            return None
        t = head + "." + (rest[:i] + "000000")[:6] + rest[i:]
    if len(t) < 6 or t[-6] not in "+-" or t[-3] != ":":
        return None  # RFC 3339 requires an offset
    try:
        return _dt.datetime.fromisoformat(t).astimezone(_dt.timezone.utc)
    except ValueError:
        return None


def _post_text(raw: str) -> Optional[PostText]:
    try:
        return PostText.parse(raw)
    except InvalidPostText:
        return None


def parse_reddit_posts(body: str, limit: int, fetched_at: _dt.datetime) -> List[SocialPost]:
    """reddit/response.rs:49-98."""
    try:
        listing = _obj(_load(body), "listing")
        if "data" not in listing:
            raise _Malformed("missing field `data`")
        children = _obj(listing["data"], "data").get("children")
        if children is None:
            children = []
        if not isinstance(children, list):
            raise _Malformed("children: expected an array")
        rows = []
        for child in children:  # the whole body is deserialised before anything is looked at
            c = _obj(child, "child")
            if "data" not in c:
                raise _Malformed("missing field `data`")
            d = _obj(c["data"], "child data")
            rows.append((_opt_str(d, "name"), _opt_str(d, "id"), _opt_str(d, "author"), _opt_str(d, "title"),
                         _opt_str(d, "selftext"), _opt_i64(d, "score"), _opt_f64(d, "created_utc")))
    except _Malformed as e:
        raise SourceFailure("reddit", "malformed response: %s" % e)
    if limit == 0:
        return []
    posts: List[SocialPost] = []
    for name, id_, author, title, selftext, score, created_utc in rows:
        pid = name if name is not None else id_  # d.name.or(d.id)
        if not pid:
            continue
        title = title or ""
        selftext = selftext or ""
        combined = title if not PostText.rust_trim(selftext) else "%s\n%s" % (title, selftext)
        text = _post_text(combined[:MAX_POST_LEN])  # chars().take(MAX_POST_LEN)
        if text is None:
            continue
        created = _timestamp(_f64_as_i64(created_utc)) if created_utc is not None else None
        posts.append(SocialPost(id=pid, source=SourceKind.REDDIT, author=author if author is not None else _UNKNOWN,
                                text=text, created_at=created if created is not None else fetched_at,
                                engagement=max(score if score is not None else 0, 0) & _U32_MAX))  # `as u32` wraps
        if len(posts) >= limit:
            break
    return posts


def parse_bluesky_posts(body: str, limit: int, fetched_at: _dt.datetime) -> List[SocialPost]:
    """bluesky/response.rs:59-114."""
    try:
        resp = _obj(_load(body), "response")
        views = resp.get("posts")
        if views is None:
            views = []
        if not isinstance(views, list):
            raise _Malformed("posts: expected an array")
        rows = []
        for view in views:
            v = _obj(view, "post view")
            author = v.get("author")
            record = v.get("record")
            author = _obj(author, "author") if author is not None else {}
            record = _obj(record, "record") if record is not None else {}
            rows.append((_opt_str(v, "uri"), _opt_str(author, "handle"), _opt_str(record, "text"),
                         _opt_str(record, "createdAt"), _opt_str(v, "indexedAt"), _opt_i64(v, "likeCount"),
                         _opt_i64(v, "repostCount"), _opt_i64(v, "replyCount")))
    except _Malformed as e:
        raise SourceFailure("bluesky", "malformed response: %s" % e)
    if limit == 0:
        return []
    posts: List[SocialPost] = []
    for uri, handle, text_raw, created_raw, indexed_raw, likes, reposts, replies in rows:
        if not uri:
            continue
        text = _post_text(text_raw or "")
        if text is None:
            continue  # empty/whitespace text -> skip, not fatal
        created = _parse_rfc3339(created_raw) or _parse_rfc3339(indexed_raw) or fetched_at
        engagement = min(sum(max(c if c is not None else 0, 0) for c in (likes, reposts, replies)), _U32_MAX)
        posts.append(SocialPost(id=uri, source=SourceKind.BLUESKY, author=handle if handle is not None else _UNKNOWN,
                                text=text, created_at=created, engagement=engagement))
        if len(posts) >= limit:
            break
    return posts
