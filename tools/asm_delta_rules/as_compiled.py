RULES = {'as compiled': lambda b: b}
