"""Writes .build_commit (git-ignored, travels to the GPU box with the snapshot — .git does not): the commit the tree was
built from, '+dirty' if tracked files differ from it.  bench.py and the profile summarisers copy it into what they write,
so every file under profiles/ names the source it was measured on."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def stamp() -> str:
    sha = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short=12", "HEAD"], capture_output=True, text=True).stdout.strip()
    dirty = subprocess.run(["git", "-C", ROOT, "status", "--porcelain", "--untracked-files=no"], capture_output=True, text=True).stdout.strip()
    s = (sha + ("+dirty" if dirty else "")) if sha else "unknown"
    with open(os.path.join(ROOT, ".build_commit"), "w") as f:
        f.write(s + "\n")
    return s


if __name__ == "__main__":
    print(stamp())
