"""Content hashes of the device sources (friedrich_amd/csrc): what a profile under profiles/ was taken with.
   source_hash.py            -> prints {"csrc_sha256": ..., "files": {name: sha256}}
tests/test_profiles_fresh.py compares the record next to the profiles with the tree (content, not mtime: a fresh checkout
gives every file the same time)."""
import hashlib
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def source_hashes(root=ROOT):
    d = os.path.join(root, "friedrich_amd", "csrc")
    files = {}
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".hpp", ".h")):
            with open(os.path.join(d, name), "rb") as f:
                files[name] = hashlib.sha256(f.read()).hexdigest()
    total = hashlib.sha256("".join(f"{k}:{v}\n" for k, v in files.items()).encode()).hexdigest()
    return {"csrc_sha256": total, "files": files}


if __name__ == "__main__":
    print(json.dumps(source_hashes(), indent=1))
