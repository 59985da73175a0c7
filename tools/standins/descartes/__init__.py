class PolygonPatch:
    pass
