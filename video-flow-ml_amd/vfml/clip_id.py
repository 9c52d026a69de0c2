"""Identity of a device-resident clip for the engine's per-frame / per-pair caches.

The caches (encoder outputs, pooled target pyramids, context maps, correlation pyramids) are keyed on
"frame i of clip X".  A device address is NOT an identity: torch's caching allocator hands a freed clip's
address to the next upload of the same size, and a fresh tensor starts at `_version == 0` again, so
`(data_ptr, _version)` keys made a second clip of the same shape reuse the first clip's features.
Every tensor OBJECT therefore gets a process-unique token the first time it is used as a clip (and
`upload_clip` assigns one straight away); `_version` stays in the key so that an in-place edit of the
clip is seen.  A new view of the same storage is a new object and simply misses the caches."""
import itertools

_next_id = itertools.count(1)


def new_id():
    return next(_next_id)


def clip_token(clip):
    """(unique id of this tensor object, its in-place version counter).  A clip that is filled frame by frame while
    the job runs (vfml.runner.ClipFeeder) carries a fixed token instead: its frames never change once uploaded, and
    the version counter moves with every upload."""
    fixed = getattr(clip, "_vfml_clip_token", None)
    if fixed is not None:
        return fixed
    tok = getattr(clip, "_vfml_clip_id", None)
    if tok is None:
        tok = next(_next_id)
        clip._vfml_clip_id = tok
    return tok, clip._version
