"""ORACLE - test infrastructure only.  In-process stand-in for pathos.multiprocessing.Pool, the only pathos
name the reference uses (rda_solver.py:9,213,725): the initializer runs once in this process (it fills the
module globals `solve_parallel` reads, rda_solver.py:268), `map` is the builtin map.  The arithmetic of the
`process_num > 1` branch (rda_solver.py:706-793) is therefore executed unchanged, without fork / pickle."""


class Pool:
    def __init__(self, processes=None, initializer=None, initargs=()):
        self.processes = processes
        if initializer is not None:
            initializer(*initargs)

    def map(self, func, iterable):
        return list(map(func, iterable))

    def close(self):
        pass

    def join(self):
        pass
