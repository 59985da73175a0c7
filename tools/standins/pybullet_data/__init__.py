def getDataPath():
    return ''
