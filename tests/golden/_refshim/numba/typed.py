class Dict(dict):
    @classmethod
    def empty(cls, key_type=None, value_type=None):
        return cls()


class List(list):
    pass
