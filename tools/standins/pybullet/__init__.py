"""empty placeholder: never called for MazeEnv"""
