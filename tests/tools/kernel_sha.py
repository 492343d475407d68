"""sha256 over the HIP sources of the product (phnet_amd/csrc/*.hip, *.h, sorted by name): what profiles/rNN_pmc_summary.json
records next to its counters, and what bench.py compares with the tree it runs in before it quotes them."""
import glob
import hashlib
import os


def kernel_sources_sha(root: str = None) -> str:
    root = root or os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    h = hashlib.sha256()
    for path in sorted(glob.glob(os.path.join(root, "phnet_amd", "csrc", "*.hip")) + glob.glob(os.path.join(root, "phnet_amd", "csrc", "*.h"))):
        h.update(os.path.basename(path).encode())
        h.update(open(path, "rb").read())
    return h.hexdigest()


if __name__ == "__main__":
    print(kernel_sources_sha())
