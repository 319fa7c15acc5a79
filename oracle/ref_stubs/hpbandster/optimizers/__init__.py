class HyperBand:
    def __init__(self, *a, **k):
        pass
