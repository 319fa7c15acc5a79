AxClient = None
