from .render import Renderer
