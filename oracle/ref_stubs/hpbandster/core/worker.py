class Worker:  # kge/job/search_grash.py subclasses this
    def __init__(self, *a, **k):
        pass
