c.NotebookApp.ip = "0.0.0.0"
c.NotebookApp.port = 9999
c.NotebookApp.open_browser = False
c.NotebookApp.allow_root = True
