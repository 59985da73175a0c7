affinity = None
