class Point:
    pass


class LineString:
    pass


class Polygon:
    pass
