"""Host-side data layer (Python 3 restatement of laed/dataset): corpus readers and batch loaders."""
