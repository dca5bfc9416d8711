"""Test stub: progress bars."""
class tqdm:
    def __init__(self, it, **k): self.it = it
    def __iter__(self): return iter(self.it)
    def set_description(self, *a, **k): pass
